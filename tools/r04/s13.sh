#!/bin/bash
cd "$GRAFT_REPO_ROOT/_old"; export TMPDIR=/tmp
O=../gpurun_out/r04o; mkdir -p $O
for i in 1 2 3 4 5 6 7 8 9 10; do
  timeout 300 python3 bench.py --steps 30 --warmup 3 --cpu-baseline 0 --secondary 0 > $O/old$i.json 2> $O/old$i.err; echo "r03 code run $i rc=$?"
done
