#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
timeout 300 python3 tools/r05/i8_stamp_probe.py 0,2,4,6 > $O/g10_i8_stamps.txt 2>&1; cat $O/g10_i8_stamps.txt
timeout 600 python3 -m pytest tests/test_gpu_int_ops.py tests/test_gpu_codec_int.py -x -q 2>&1 | tail -3
