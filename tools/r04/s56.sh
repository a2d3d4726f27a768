#!/bin/bash
# GPU session 56: narrow layers on large maps: natural (Morton) order against neighbour-pattern order (gather-bound vs MFMA-bound)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
for sh in "32 32" "48 32" "16 32" "64 64"; do
  set -- $sh
  timeout 200 python3 tools/conv_probe.py 0 $1 $2 10 2>&1 | grep "level"
done
