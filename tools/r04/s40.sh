#!/bin/bash
# GPU session 40: a stream per frame context, with the current pipeline (named stages, shared parameter objects)
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
timeout 300 python3 bench.py --cpu-baseline 0 --secondary 0 --steps 20 --warmup 5 > $O/o0.json 2> $O/o0.err; show $O/o0.json "one stream"
timeout 300 python3 bench.py --cpu-baseline 0 --secondary 0 --steps 20 --warmup 5 --own-streams 1 > $O/o1.json 2> $O/o1.err; show $O/o1.json "own streams"
timeout 300 python3 bench.py --cpu-baseline 0 --secondary 0 --steps 20 --warmup 5 --own-streams 1 --frames-in-flight 3 > $O/o2.json 2> $O/o2.err; show $O/o2.json "own streams, 3 frames"
tail -3 $O/o1.err
