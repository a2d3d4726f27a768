#!/bin/bash
# GPU session 58: occupancy predictors on a second stream beside the feature chain (FPCC_OCCUPANCY_STREAM=1): bytes, determinism, bench A/B
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r04I; mkdir -p $O
FPCC_OCCUPANCY_STREAM=1 timeout 900 python3 -m pytest tests/test_gpu_codec_v2.py tests/test_gpu_serving.py -x -q 2>&1 | tail -3
FPCC_OCCUPANCY_STREAM=1 timeout 300 python3 tools/r04/stress2.py 200 1024 2>&1 | tail -2
FPCC_OCCUPANCY_STREAM=1 timeout 300 python3 tools/r04/stress3.py 100 1024 2>&1 | tail -1
for rep in 1 2 3; do
for f in 0 1; do
  for d in 2 1; do
  FPCC_OCCUPANCY_STREAM=$f timeout 300 python3 bench.py --cpu-baseline 0 --secondary 0 --steps 20 --warmup 5 --frames-in-flight $d > $O/q.json 2> $O/q.err
  python3 - <<PY
import json
d=json.loads(open('$O/q.json').read().strip().splitlines()[-1]); print('occupancy_stream=$f depth $d', d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_step'], d['config']['encode_ms'], d['config']['decode_ms'])
PY
  done
done; done
