#!/bin/bash
# GPU session 36: named stages (one frame encodes while the other decodes): default bench x3, no-baseline x3
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r04H; mkdir -p $O
export FPCC_BENCH_STEP_TIMES=1
show() { python3 - <<PY
import json
d=json.loads(open('$1').read().strip().splitlines()[-1]); r=d['roofline']; c=d['config']
print('$2', d['value'], d['ms_per_step'], 'frac', r['frac'], 'kernel_ms', r['kernel_ms_per_step'])
PY
grep "step completions" ${1%.json}.err
}
for i in 1 2 3; do
timeout 600 python3 bench.py --cpu-baseline 0 --secondary 0 > $O/c$i.json 2> $O/c$i.err; show $O/c$i.json "neither"
done
timeout 600 python3 bench.py > $O/d.json 2> $O/d.err; show $O/d.json "default"
timeout 600 python3 bench.py --steps 20 --warmup 5 > $O/d2.json 2> $O/d2.err; show $O/d2.json "default 20 5"
timeout 600 python3 bench.py --steps 20 --warmup 5 --cpu-baseline 0 --secondary 0 > $O/d3.json 2> $O/d3.err; show $O/d3.json "neither 20 5"
