#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
timeout 300 python3 tools/r05/int_phases.py > $O/g26_phases.txt 2>&1
timeout 300 python3 tools/r05/int_phases.py sync > $O/g26_phases_sync.txt 2>&1
cat $O/g26_phases.txt $O/g26_phases_sync.txt
