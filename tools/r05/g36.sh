#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
timeout 600 python3 tools/r05/train_ops_by_line.py > $O/g36_train_ops_by_line.txt 2>&1; tail -3 $O/g36_train_ops_by_line.txt
