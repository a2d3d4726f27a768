#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
timeout 1200 python3 -m pytest tests/test_gpu_autograd.py tests/test_gpu_training.py tests/test_gpu_train_v3.py tests/test_gpu_lossl_float.py tests/test_gpu_entropy_kernel.py tests/test_gpu_entropy_glue.py -x -q 2>&1 | tail -6
timeout 600 python3 tools/r05/train_ops_by_line.py > $O/g37_train_ops_by_line.txt 2>&1; head -3 $O/g37_train_ops_by_line.txt
timeout 600 python3 bench_train.py --steps 10 --warmup 3 2>&1 | tail -2
BY_TIME=1 TOP=14 timeout 600 python3 tools/train_launches.py 2>&1 | grep -v Warning | head -18
TOP=8 timeout 300 python3 tools/int_launches.py 2>&1 | grep -v Warning | grep -v amdgpu
