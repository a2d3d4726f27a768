#!/bin/bash
# GPU session 35: which default flag costs the pipelined timed region its throughput?
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
timeout 600 python3 bench.py --cpu-baseline 1 --secondary 0 > $O/a.json 2> $O/a.err; show $O/a.json "baseline only"
timeout 600 python3 bench.py --cpu-baseline 0 --secondary 1 > $O/b.json 2> $O/b.err; show $O/b.json "secondary only"
timeout 600 python3 bench.py --cpu-baseline 0 --secondary 0 > $O/c.json 2> $O/c.err; show $O/c.json "neither"
timeout 600 python3 bench.py > $O/d.json 2> $O/d.err; show $O/d.json "both"
timeout 600 python3 bench.py --cpu-baseline 0 --secondary 0 > $O/e.json 2> $O/e.err; show $O/e.json "neither"
