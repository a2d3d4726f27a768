#!/bin/bash
O=gpurun_out/r03_fold; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_fullsize.py tests/test_gpu_codec_v2.py -x -q -m gpu 2>&1 | tail -3 | tee $O/tests2.txt
