#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r02i; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_conv.py -x -q 2>&1 | tail -3
run() { timeout 300 python tools/conv_probe.py "$@" 2>&1 | grep -v amdgpu.ids; }
ONLY=pattern NBWS=1,2 RINGS=1,2,3 run 2 128 128 20 | tee $O/l2.txt
ONLY=pattern NBWS=1,2 RINGS=1,2,3 run 3 128 128 20 | tee $O/l3.txt
ONLY=pattern NBWS=1,2 RINGS=1,2,3 run 3 256 128 20 | tee $O/l3_256.txt
ONLY=pattern NBWS=2 RINGS=1,2 run 1 128 128 20 | tee $O/l1.txt
for lvl in 4 5; do
  ONLY=natural NBWS=0 run $lvl 128 128 30 | tee $O/l${lvl}_split.txt
  FPCC_SPLIT_MAX_ROWS=0 ONLY=natural NBWS=1 RINGS=1,2,3 run $lvl 128 128 30 | tee $O/l${lvl}_wave.txt
done
