#!/bin/bash
export TMPDIR=/tmp
for c in e11f42a b05001d; do
  cd "$GRAFT_REPO_ROOT/_w_$c"; O=../gpurun_out/r04p; mkdir -p $O; fails=0
  for i in 1 2 3 4 5 6 7 8 9 10; do
    timeout 300 python3 bench.py --steps 30 --warmup 3 --cpu-baseline 0 --secondary 0 > $O/$c.$i.json 2> $O/$c.$i.err || fails=$((fails+1))
  done
  echo "commit $c: $fails failures of 10"
done
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04p; fails=0
for i in 1 2 3 4 5 6 7 8 9 10; do
  FPCC_NBR_ROWS=0 timeout 300 python3 bench.py --steps 30 --warmup 3 --cpu-baseline 0 --secondary 0 --frames-in-flight 1 > $O/head.$i.json 2> $O/head.$i.err || fails=$((fails+1))
done
echo "HEAD (+working tree), offset-major tables: $fails failures of 10"
