#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
for cfg in "8 2 1" "8 2 0" "8 3 0" "8 4 0" "4 3 0" "4 4 0"; do set -- $cfg
  python3 bench.py --steps 12 --warmup 3 --batch $1 --frames-in-flight $2 --stages $3 --secondary 0 --cpu-baseline 0 > $O/g17_b$1_d$2_s$3.json 2> $O/g17_b$1_d$2_s$3.err
  python3 -c "
import json
d = json.loads(open('$O/g17_b$1_d$2_s$3.json').read().strip().splitlines()[-1])
print('batch $1 depth $2 stages $3: value', d['value'], 'ms/step', d['ms_per_step'], 'frac', d['roofline']['frac'])" || tail -3 $O/g17_b$1_d$2_s$3.err
done
