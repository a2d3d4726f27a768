#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r02tr; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_autograd.py tests/test_gpu_training.py tests/test_gpu_train_v3.py -x -q > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt
tail -4 $O/pytest.txt
for p in fresh off fresh off; do
  FPCC_TRAIN_PACK=$p timeout 300 python bench_train.py --steps 12 --warmup 4 2>/dev/null | tail -1 | python -c "
import sys, json
d=json.loads(sys.stdin.read()); print('$p', d.get('ms_per_step'), d.get('value'))"
done
