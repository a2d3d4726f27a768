#!/bin/bash
# GPU session 4 (round 4): ablations of k_conv_lds (timing only): which of fragment reads / DMAs / barrier holds the MFMA pipe at 69 %
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r04d; mkdir -p $O
ONLY=pattern LDS=2,4 DBG=0,4,8,16,24,28 timeout 600 python3 tools/conv_probe.py 1 128 128 20 > $O/ablate.txt 2>&1
ONLY=pattern LDS=2 DBG=0,4,8,16,24,28 timeout 600 python3 tools/conv_probe.py 2 128 128 20 >> $O/ablate.txt 2>&1
grep -v amdgpu $O/ablate.txt
