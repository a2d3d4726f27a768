#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
timeout 600 python3 tools/r05/int_ops_by_line.py > $O/g22_int_ops_by_line.txt 2>&1; tail -5 $O/g22_int_ops_by_line.txt
