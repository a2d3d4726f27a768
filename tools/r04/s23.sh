#!/bin/bash
# GPU session 23: colour chain-order parity; integer codec baseline (per-shape table, launch census)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r04y; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_codec_color.py -q > $O/color.txt 2>&1; tail -4 $O/color.txt
timeout 300 python3 tools/timeline_int.py > $O/int_timeline.txt 2>&1; tail -3 $O/int_timeline.txt
timeout 300 python3 tools/int_trace.py > $O/int_trace.txt 2>&1; cat $O/int_trace.txt | head -40
TOP=25 timeout 300 python3 tools/int_launches.py > $O/int_launches.txt 2>&1; cat $O/int_launches.txt
