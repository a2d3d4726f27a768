#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r02c; mkdir -p $O
for lvl in 1 2 3; do
  NBWS=1,2,4 DBG=0,1,2,3 ONLY=pattern timeout 300 python tools/conv_probe.py $lvl 128 128 20 2>&1 | grep -v amdgpu.ids | tee $O/dbg_l$lvl.txt
  NBWS=2,4 DBG=0,1,2,3 ONLY=natural timeout 300 python tools/conv_probe.py $lvl 128 128 20 2>&1 | grep -v amdgpu.ids | tee $O/dbg_nat_l$lvl.txt
done
