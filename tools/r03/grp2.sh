#!/bin/bash
# grouped evaluation against the order-1 unit once more, every series behind 60 ms of warm-up (clock at 2.4 GHz for all variants)
O=gpurun_out/r03_grp2; mkdir -p $O; rm -f $O/probe.txt
export FPCC_EXPERIMENT=1
for lvl in 1 2 3 4 5; do
  for shape in "128 128" "64 64"; do
    ONLY=pattern GROUPED=1 timeout 200 python tools/conv_probe.py $lvl $shape 20 2>&1 | grep -v amdgpu.ids | tee -a $O/probe.txt
  done
done
