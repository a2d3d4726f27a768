#!/bin/bash
O=gpurun_out/r03_int2; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_codec_int.py tests/test_gpu_int_ops.py tests/test_gpu_ptq.py tests/test_gpu_lossl_float.py tests/test_gpu_fullsize.py -x -q -m gpu > $O/pytest.txt 2>&1; echo "pytest rc $?" >> $O/pytest.txt
tail -15 $O/pytest.txt
TOP=12 timeout 300 python tools/int_launches.py 2>&1 | grep -v "amdgpu.ids\|Warning\|warn" > $O/launches.txt; cat $O/launches.txt
timeout 300 python tools/timeline_int.py 2>&1 | grep -v amdgpu.ids | tee $O/timeline.txt
