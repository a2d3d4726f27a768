#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
for w in 0 1 2 3 4 6; do FPCC_HOST_WARMERS=$w python3 tools/r05/dec_bench.py 300000; done 2>&1 | tee $O/g13_dec_bench.txt
for w in 0 2 3 4; do echo "== FPCC_HOST_WARMERS=$w"; FPCC_HOST_WARMERS=$w TOP=3 python3 tools/int_launches.py 2>&1 | grep -E "^=="; done | tee $O/g13_int.txt
timeout 600 python3 -m pytest tests/test_gpu_codec_int.py tests/test_gpu_autograd.py tests/test_gpu_codec_many.py -x -q 2>&1 | tail -3
