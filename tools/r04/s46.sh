#!/bin/bash
# GPU session 46: where the host time of a frame goes (cProfile, one frame at a time)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r04I; mkdir -p $O
timeout 300 python3 tools/pyprofile.py > $O/pyprofile.txt 2>&1; head -60 $O/pyprofile.txt
