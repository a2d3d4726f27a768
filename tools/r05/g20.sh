#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
timeout 900 python3 -m pytest tests/test_gpu_autograd.py tests/test_gpu_training.py tests/test_gpu_train_v3.py -x -q 2>&1 | tail -4
for i in 1 2; do timeout 300 python3 bench_train.py --steps 8 --warmup 3 2>/dev/null | head -c 260; echo; done
TOP=14 BY_TIME=1 timeout 300 python3 tools/train_launches.py 2>&1 | grep -E "^wall|k_conv_valu|k_wgrad_small|k_wgrad" | head -12
