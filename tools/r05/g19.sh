#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
for i in 1 2 3; do TOP=1 python3 tools/int_launches.py 2>&1 | grep -E "^=="; done
timeout 600 python3 -m pytest tests/test_gpu_codec_int.py -x -q 2>&1 | tail -2
