#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r02i; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_int_ops.py tests/test_gpu_codec_int.py tests/test_gpu_ptq.py tests/test_gpu_codec_v3.py tests/test_gpu_lossl_float.py tests/test_gpu_train_v3.py -x -q > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt
tail -3 $O/pytest.txt
timeout 120 python tools/timeline_int.py 2>&1 | tail -3
