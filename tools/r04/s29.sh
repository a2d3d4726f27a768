#!/bin/bash
# GPU session 29: lean requantisers in the stand-alone epilogue and the int8 copies; int tests; launch census
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r04z; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_int_ops.py tests/test_gpu_codec_int.py tests/test_gpu_lossl_float.py -x -q > $O/int_tests.txt 2>&1; tail -3 $O/int_tests.txt
TOP=8 timeout 300 python3 tools/int_launches.py > $O/int_launches3.txt 2>&1; grep -v Warn $O/int_launches3.txt | head -24
timeout 300 python3 tools/timeline_int.py > $O/int_timeline4.txt 2>&1; tail -3 $O/int_timeline4.txt
