#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r02v; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_codec_v3.py -x -q --durations=8 > $O/pytest_v3.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_v3.txt
tail -40 $O/pytest_v3.txt
