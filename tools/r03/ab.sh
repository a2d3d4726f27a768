#!/bin/bash
# same-box A/B of two builds of libfpcc_hip.so: the tree's ("new") and fastpcc_amd/csrc/alt/ ("old": built from another source state, not committed)
O=gpurun_out/r03_ab; mkdir -p $O; rm -f $O/*.txt
run() {
  for lvl in ${LEVELS:-1 2 3}; do
    for shape in "128 128" "64 64"; do
      ONLY=pattern timeout 200 python tools/conv_probe.py $lvl $shape 30 2>&1 | grep -v amdgpu.ids | sed "s/^/$1 /" | tee -a $O/probe.txt
    done
  done
}
cp fastpcc_amd/csrc/libfpcc_hip.so /tmp/new.so
timeout 600 python -m pytest tests/test_gpu_conv.py -x -q -m gpu 2>&1 | tail -2
run new
cp fastpcc_amd/csrc/alt/libfpcc_hip.so fastpcc_amd/csrc/libfpcc_hip.so; run old
cp /tmp/new.so fastpcc_amd/csrc/libfpcc_hip.so; run new
cp fastpcc_amd/csrc/alt/libfpcc_hip.so fastpcc_amd/csrc/libfpcc_hip.so; run old
