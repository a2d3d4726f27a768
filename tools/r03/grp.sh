#!/bin/bash
O=gpurun_out/r03_grp; mkdir -p $O; rm -f $O/probe.txt
export FPCC_EXPERIMENT=1
for lvl in 1 2 3 4 5 6; do
  for shape in "128 128" "256 128" "64 64"; do
    ONLY=pattern GROUPED=1 timeout 200 python tools/conv_probe.py $lvl $shape 20 2>&1 | grep -v amdgpu.ids | tee -a $O/probe.txt
  done
done
