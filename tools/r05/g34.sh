#!/bin/bash
# batch size sweep of the default bench beyond 8
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
for b in ${BATCHES:-8 12 16}; do
  timeout 900 python3 bench.py --batch $b --cpu-baseline 0 --secondary 0 > $O/g34_b$b.json 2> $O/g34_b$b.err
  python3 - $O/g34_b$b.json $b <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print('batch', sys.argv[2], 'value', d['value'], 'one_frame', d.get('value_one_frame'), 'frac', d['roofline']['frac'], 'ms_per_step', d['ms_per_step'])
except Exception as e:
    print('batch', sys.argv[2], 'failed', e)
PY
done
tail -3 $O/g34_b16.err
