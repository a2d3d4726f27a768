#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
for w in 0 1 2 4; do echo "warmers $w"; FPCC_HOST_WARMERS=$w timeout 300 python3 tools/timeline_int.py 2>&1 | tail -3; done
nproc; lscpu | grep -i "model name\|L3\|L2" 
