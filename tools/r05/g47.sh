#!/bin/bash
# row-warming helper thread with the process bound to the GPU's NUMA node (the default now): alternating runs on one box
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
for i in 1 2 3 4 5; do for w in 0 1; do echo "warmers $w: $(FPCC_HOST_WARMERS=$w timeout 300 python3 tools/timeline_int.py 2>&1 | tail -3 | tr '\n' ' ' | sed 's/113108 voxels://g; s/bytes [0-9]* bpp [0-9.]* lossless True//g')"; done; done
