#!/bin/bash
# GPU session 32: is a 20-step timed region slower per step than a 10-step one?  per-step completion times
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r04G; mkdir -p $O
export FPCC_BENCH_STEP_TIMES=1
for cfg in "20 5" "10 3" "20 5" "40 5" "10 3"; do
  set -- $cfg
  timeout 300 python3 bench.py --steps $1 --warmup $2 --cpu-baseline 0 --secondary 0 > $O/c.json 2> $O/c.err
  python3 - <<PY
import json
d=json.loads(open('$O/c.json').read().strip().splitlines()[-1]); r=d['roofline']
print('steps $1', d['value'], d['ms_per_step'], 'frac', r['frac'], 'kernel_ms', r['kernel_ms_per_step'])
PY
  grep "step completions" $O/c.err
done
