#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r04n; mkdir -p $O
for i in 1 2 3 4 5; do
  timeout 300 python3 bench.py --steps 30 --warmup 3 --cpu-baseline 0 --secondary 0 --frames-in-flight 1 > $O/a$i.json 2> $O/a$i.err; echo "rows=1 run $i rc=$?"
done
for i in 1 2 3; do
  FPCC_NBR_ROWS=0 timeout 300 python3 bench.py --steps 30 --warmup 3 --cpu-baseline 0 --secondary 0 --frames-in-flight 1 > $O/b$i.json 2> $O/b$i.err; echo "rows=0 run $i rc=$?"
done
