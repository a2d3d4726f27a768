#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r05
for cfg in "4 3" "4 4" "6 2" "8 2" "8 3" "2 3" "2 4"; do
  set -- $cfg
  python bench.py --steps 12 --warmup 3 --batch $1 --frames-in-flight $2 --secondary 0 --cpu-baseline 0 > gpurun_out/r05/g4_bench_b$1_d$2.json 2> gpurun_out/r05/g4_bench_b$1_d$2.err
  python - <<PY
import json
try:
    d = json.loads(open('gpurun_out/r05/g4_bench_b$1_d$2.json').read().strip().splitlines()[-1])
    print('batch $1 depth $2: value', d['value'], 'one_frame', d['value_one_frame'], 'ms/step', d['ms_per_step'], 'frac', d['roofline']['frac'], 'kernel ms', d['roofline']['kernel_ms_per_step'])
except Exception as e:
    print('batch $1 depth $2: failed', e)
PY
done 2>&1 | tee gpurun_out/r05/g4_sweep.log
