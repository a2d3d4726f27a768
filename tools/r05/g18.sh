#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
python3 tools/pyprofile.py > $O/g18_pyprofile.txt 2>&1; head -60 $O/g18_pyprofile.txt
