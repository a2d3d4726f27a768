#!/bin/bash
# same-box A/B of two builds of libfpcc_hip.so: the tree's and fastpcc_amd/csrc/alt/ (built from another source state, not committed)
O=gpurun_out/r03_ab; mkdir -p $O; rm -f $O/*.txt
run() {
  for lvl in 1 2 3 4 5; do
    ONLY=pattern timeout 200 python tools/conv_probe.py $lvl 128 128 30 2>&1 | grep -v amdgpu.ids | sed "s/^/$1 /" | tee -a $O/probe.txt
  done
}
run new
cp fastpcc_amd/csrc/libfpcc_hip.so /tmp/new.so; cp fastpcc_amd/csrc/alt/libfpcc_hip.so fastpcc_amd/csrc/libfpcc_hip.so
run old
cp /tmp/new.so fastpcc_amd/csrc/libfpcc_hip.so
run new
cp fastpcc_amd/csrc/alt/libfpcc_hip.so fastpcc_amd/csrc/libfpcc_hip.so
run old
