#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_codec_int.py -x -q 2>&1 | tail -15
timeout 600 python3 tools/r05/int_many_bench.py 5 1,2,4,8 2>&1 | tee $O/g21_int_many.txt | tail -8
