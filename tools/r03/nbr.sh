#!/bin/bash
O=gpurun_out/r03_nbr; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_coords.py tests/test_gpu_conv.py -x -q -m gpu 2>&1 | tail -2
echo new; timeout 200 python tools/nbr27_probe.py 2>&1 | grep -v amdgpu | tee $O/new.txt
cp fastpcc_amd/csrc/libfpcc_hip.so /tmp/new.so; cp fastpcc_amd/csrc/alt/libfpcc_hip.so fastpcc_amd/csrc/libfpcc_hip.so
echo old; timeout 200 python tools/nbr27_probe.py 2>&1 | grep -v amdgpu | tee $O/old.txt
cp /tmp/new.so fastpcc_amd/csrc/libfpcc_hip.so
