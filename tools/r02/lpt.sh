#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r02l; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_fullsize.py tests/test_gpu_codec_int.py -x -q > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt
tail -4 $O/pytest.txt
for i in 1 2 3; do timeout 200 python bench.py --steps 10 --warmup 3 --cpu-baseline 0 --secondary 0 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config']['encode_ms'], d['config']['decode_ms'], d['roofline']['kernel_ms_per_step'], d['roofline']['frac'], d['config']['bytes'])"; done
