#!/bin/bash
O=gpurun_out/r03_gapramp; mkdir -p $O
timeout 300 python tools/gap_ramp_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/gapramp.txt
