#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r04G; mkdir -p $O
timeout 600 python3 -m pytest tests/test_gpu_serving.py -q > $O/serving.txt 2>&1; tail -15 $O/serving.txt
