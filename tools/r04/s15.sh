#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r04q; mkdir -p $O
timeout 600 python3 tools/r04/stress.py 200 > $O/stress_rows1.txt 2>&1; tail -5 $O/stress_rows1.txt
FPCC_NBR_ROWS=0 timeout 600 python3 tools/r04/stress.py 200 > $O/stress_rows0.txt 2>&1; tail -5 $O/stress_rows0.txt
