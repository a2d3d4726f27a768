#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
timeout 600 python3 -m pytest tests/test_gpu_entropy_glue.py -q 2>&1 | tail -4
