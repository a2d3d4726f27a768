#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
timeout 600 python3 tools/r04/copy_sources.py 2>&1 | grep -v "amdgpu.ids\|Warn\|warn" | tail -45
