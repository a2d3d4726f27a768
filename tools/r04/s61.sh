#!/bin/bash
# GPU session 61: unusual step counts of the bench (1, 2, 3, 9) with the default depth
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r04I; mkdir -p $O
for cfg in "1 0" "2 1" "3 1" "9 2"; do
  set -- $cfg
  timeout 300 python3 bench.py --cpu-baseline 0 --secondary 0 --steps $1 --warmup $2 > $O/u.json 2> $O/u.err; rc=$?
  python3 - <<PY
import json
try:
    d=json.loads(open('$O/u.json').read().strip().splitlines()[-1]); print('steps $1 warmup $2 rc=$rc', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['launches_per_step'], d['roofline']['event_traced_steps'][:12])
except Exception as e:
    print('steps $1 warmup $2 rc=$rc FAILED', e); print(open('$O/u.err').read()[-800:])
PY
done
