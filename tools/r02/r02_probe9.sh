#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r02k; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_conv.py -x -q 2>&1 | tail -3
run() { timeout 300 python tools/conv_probe.py "$@" 2>&1 | grep -v amdgpu.ids; }
ONLY=pattern NBWS=0,2,-2,-3 DBG=0 run 1 128 128 20 | tee $O/l1.txt
ONLY=pattern NBWS=-2,-3 DBG=3,4 run 1 128 128 20 | tee -a $O/l1.txt
ONLY=pattern NBWS=0,2,-2,-3 run 1 64 64 20 | tee $O/l1_64.txt
ONLY=pattern NBWS=0,1,2,-2,-3 run 2 128 128 20 | tee $O/l2.txt
ONLY=pattern NBWS=2,-2,-3 run 2 256 128 20 | tee $O/l2_256.txt
ONLY=pattern NBWS=0,1,-2,-3 run 3 128 128 20 | tee $O/l3.txt
