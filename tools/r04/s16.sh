#!/bin/bash
# GPU session 16: strict chain-order parity test, the two-rank bench dry run, the int / colour / v3 suites after today's kernel changes
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r04r; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_codec_v2.py -x -q -k "chain_order" > $O/chain.txt 2>&1; tail -15 $O/chain.txt
timeout 900 python3 -m pytest tests/test_gpu_bench_ddp.py -x -q > $O/ddp.txt 2>&1; tail -15 $O/ddp.txt
timeout 1500 python3 -m pytest tests/test_gpu_int_ops.py tests/test_gpu_codec_int.py tests/test_gpu_autograd.py -x -q > $O/int.txt 2>&1; tail -5 $O/int.txt
