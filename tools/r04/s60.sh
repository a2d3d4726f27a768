#!/bin/bash
# GPU session 60: the encode stage left early (host coding tail outside the stage): A/B, alternating repeats; stress for parity
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r04I; mkdir -p $O
timeout 300 python3 -m pytest tests/test_gpu_serving.py tests/test_gpu_bench_ddp.py -q 2>&1 | tail -2
for rep in 1 2 3 4; do
for e in 1 0; do
  FPCC_BENCH_EARLY_RELEASE=$e timeout 300 python3 bench.py --cpu-baseline 0 --secondary 0 --steps 20 --warmup 5 > $O/e.json 2> $O/e.err
  python3 - <<PY
import json
d=json.loads(open('$O/e.json').read().strip().splitlines()[-1]); print('early_release=$e', d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_step'])
PY
done; done
