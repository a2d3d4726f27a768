#!/bin/bash
O=gpurun_out/r03_int3; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_codec_int.py tests/test_gpu_lossl_float.py tests/test_gpu_codec_v3.py tests/test_gpu_train_v3.py tests/test_gpu_ptq.py tests/test_gpu_fullsize.py -x -q -m gpu > $O/pytest.txt 2>&1; echo "pytest rc $?" >> $O/pytest.txt
tail -6 $O/pytest.txt
