#!/bin/bash
# GPU session 41: three contexts with named stages (both stages always occupied?) against two
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r04I; mkdir -p $O
export FPCC_BENCH_STEP_TIMES=1
show() { python3 - <<PY
import json
d=json.loads(open('$1').read().strip().splitlines()[-1]); r=d['roofline']; c=d['config']
print('$2', d['value'], d['ms_per_step'], 'frac', r['frac'], 'kernel_ms', r['kernel_ms_per_step'])
PY
grep "step completions" ${1%.json}.err
}
for d in 2 3 2 3; do
timeout 300 python3 bench.py --cpu-baseline 0 --secondary 0 --steps 20 --warmup 5 --frames-in-flight $d > $O/f$d.json 2> $O/f$d.err; show $O/f$d.json "depth $d"
done
