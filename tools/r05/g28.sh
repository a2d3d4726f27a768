#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_codec_int.py -x -q 2>&1 | tail -5
timeout 300 python3 tools/r05/int_phases.py > $O/g28_phases.txt 2>&1
timeout 300 python3 tools/r05/int_phases.py sync > $O/g28_phases_sync.txt 2>&1
cat $O/g28_phases.txt $O/g28_phases_sync.txt
