#!/bin/bash
# GPU session 27: integer codec after the epilogue / prologue work: timeline, per-shape table, launch census
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r04z; mkdir -p $O
timeout 300 python3 tools/timeline_int.py > $O/int_timeline2.txt 2>&1; tail -2 $O/int_timeline2.txt
timeout 300 python3 tools/int_trace.py > $O/int_trace2.txt 2>&1; head -14 $O/int_trace2.txt
TOP=12 timeout 300 python3 tools/int_launches.py > $O/int_launches2.txt 2>&1; grep -v Warn $O/int_launches2.txt | head -40
