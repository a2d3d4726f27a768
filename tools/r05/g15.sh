#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
timeout 900 python3 -m pytest tests/test_gpu_codec_many.py tests/test_gpu_codec_v2.py tests/test_gpu_serving.py -q -W error::UserWarning 2>&1 | tail -40
