#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r02t; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_train_v3.py tests/test_gpu_lossl_float.py tests/test_gpu_ptq.py -x -q --durations=5 > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt
tail -60 $O/pytest.txt
