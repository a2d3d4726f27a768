#!/bin/bash
# GPU session 21: alternating-frame race detector, several fresh processes
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r04w; mkdir -p $O
for i in 1 2 3 4; do timeout 600 python3 tools/r04/stress2.py 250 1024 > $O/stress2_$i.txt 2>&1; tail -4 $O/stress2_$i.txt; done
