#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r02d; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_conv.py -x -q > $O/pytest_conv.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_conv.txt
tail -4 $O/pytest_conv.txt
run() { timeout 300 python tools/conv_probe.py "$@" 2>&1 | grep -v amdgpu.ids; }
ONLY=pattern NBWS=2,4 DEPTHS=1,2 run 1 128 128 20 | tee $O/l1.txt
ONLY=pattern NBWS=1,2 DEPTHS=1,2,3,4 run 2 128 128 20 | tee $O/l2.txt
ONLY=pattern NBWS=1,2 DEPTHS=1,2,3,4 run 2 256 128 20 | tee $O/l2_256.txt
ONLY=pattern NBWS=1,2 DEPTHS=1,2,3,4 run 3 128 128 20 | tee $O/l3.txt
ONLY=pattern NBWS=1,2 DEPTHS=1,2 run 1 64 64 20 | tee $O/l1_64.txt
for lvl in 4 5 6; do
  ONLY=natural NBWS=0 run $lvl 128 128 30 | tee $O/l${lvl}_split.txt
  FPCC_SPLIT_MAX_ROWS=0 ONLY=natural NBWS=1,2 DEPTHS=1,2,4 run $lvl 128 128 30 | tee $O/l${lvl}_wave.txt
  FPCC_SPLIT_MAX_ROWS=0 ONLY=natural NBWS=1 DEPTHS=2,4 run $lvl 256 128 30 | tee $O/l${lvl}_wave256.txt
done
