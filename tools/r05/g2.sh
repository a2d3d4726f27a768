#!/bin/bash
# second GPU session: colour batches + fallback test, then bench.py over (batch, frames in flight)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r05
python -m pytest tests/test_gpu_codec_many.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r05/g2_many.log
for cfg in "1 1" "1 2" "2 1" "2 2" "4 1" "4 2" "3 2"; do
  set -- $cfg
  python bench.py --steps 12 --warmup 3 --batch $1 --frames-in-flight $2 --secondary 0 --cpu-baseline 0 > gpurun_out/r05/g2_bench_b$1_d$2.json 2> gpurun_out/r05/g2_bench_b$1_d$2.err
  python - <<PY
import json
try:
    d = json.loads(open('gpurun_out/r05/g2_bench_b$1_d$2.json').read().strip().splitlines()[-1])
    print('batch $1 depth $2: value', d['value'], 'one_frame', d['value_one_frame'], 'ms/step', d['ms_per_step'], 'frac', d['roofline']['frac'], 'kernel ms', d['roofline']['kernel_ms_per_step'], 'retries', d['config']['coder_handover_retries'])
except Exception as e:
    print('batch $1 depth $2: failed', e)
PY
done > gpurun_out/r05/g2_sweep.log 2>&1
cat gpurun_out/r05/g2_many.log gpurun_out/r05/g2_sweep.log
