#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
timeout 900 python3 -m pytest tests/test_gpu_codec_int.py -x -q 2>&1 | tail -15
