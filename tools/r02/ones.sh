#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r02o; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_codec_v2.py tests/test_gpu_fullsize.py tests/test_gpu_codec_color.py tests/test_gpu_training.py -x -q > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt
tail -5 $O/pytest.txt
