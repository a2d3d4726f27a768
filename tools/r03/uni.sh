#!/bin/bash
# A/B of the wave-uniform loop control in k_conv_wave: the same probe lines as tools/r03/grp.sh plus the conv parity tests
O=gpurun_out/r03_uni; mkdir -p $O; rm -f $O/probe.txt
export FPCC_EXPERIMENT=1
for lvl in 1 2 3 4; do
  for shape in "128 128" "256 128" "64 64"; do
    ONLY=pattern GROUPED=1 timeout 200 python tools/conv_probe.py $lvl $shape 20 2>&1 | grep -v amdgpu.ids | tee -a $O/probe.txt
  done
done
timeout 900 python -m pytest tests/test_gpu_conv.py -x -q -m gpu 2>&1 | tail -3 | tee $O/tests.txt
timeout 300 python bench.py 2>&1 | tail -1 | tee $O/bench.json
