#!/bin/bash
# GPU session 55: pipelined frames at bench scale against the single-frame code (bytes and decoded clouds), 200 frames of four kinds
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
timeout 900 python3 tools/r04/stress3.py 200 1024 2>&1 | grep -v amdgpu.ids | tail -4
