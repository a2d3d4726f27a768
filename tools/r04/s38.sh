#!/bin/bash
# GPU session 38: which form of the order-3 evaluation per level, after this round's prologue / epilogue changes (thresholds were set in round 3)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r04I; mkdir -p $O
for lvl in 1 2 3 4; do
  for sh in "128 128" "64 64" "256 128"; do
    set -- $sh
    FORMS=1 ONLY=pattern timeout 200 python3 tools/conv_probe.py $lvl $1 $2 20 2>/dev/null | grep "level"
  done
done | tee $O/forms.txt
