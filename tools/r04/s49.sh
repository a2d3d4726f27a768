#!/bin/bash
# GPU session 49: switch interval A/B with repeats (alternating), one box
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r04I; mkdir -p $O
for rep in 1 2 3 4; do
for sw in 2e-4 1e-4 5e-5; do
  FPCC_SWITCH_INTERVAL=$sw timeout 300 python3 bench.py --cpu-baseline 0 --secondary 0 --steps 20 --warmup 5 > $O/sw.json 2> $O/sw.err
  python3 - <<PY
import json
d=json.loads(open('$O/sw.json').read().strip().splitlines()[-1]); print('$sw', d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_step'])
PY
done; done
