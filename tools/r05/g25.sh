#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_codec_int.py tests/test_gpu_int_ops.py -x -q 2>&1 | tail -15
timeout 300 python3 tools/timeline_int.py 2>&1 | tail -5
