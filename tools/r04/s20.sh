#!/bin/bash
# GPU session 20: poisoned heap (uninitialised reads?)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r04v; mkdir -p $O
timeout 900 python3 tools/r04/poison.py 1024 2 > $O/poison.txt 2>&1; tail -25 $O/poison.txt
