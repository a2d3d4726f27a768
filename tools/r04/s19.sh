#!/bin/bash
# GPU session 19: position-ordered table fix; whole GPU suite
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r04u; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_codec_color.py tests/test_gpu_conv.py -q -k "color or position_ordered or row_major" > $O/color.txt 2>&1; tail -6 $O/color.txt
timeout 2400 python3 -m pytest tests -m gpu -q > $O/gpu_suite.txt 2>&1; tail -12 $O/gpu_suite.txt
