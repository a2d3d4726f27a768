#!/bin/bash
# GPU session 34: events recorded inside the C call; CPU baseline released after the GPU legs: default bench twice, no-baseline once, depth 1 once
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r04H; mkdir -p $O
show() { python3 - <<PY
import json
d=json.loads(open('$1').read().strip().splitlines()[-1]); r=d['roofline']; c=d['config']
print('$2', d['value'], d['ms_per_step'], 'enc', c['encode_ms'], 'dec', c['decode_ms'], 'frac', r['frac'], 'kernel_ms', r['kernel_ms_per_step'], 'launches', r['launches_per_step'], 'clock', r['shader_clock_mhz'], 'cpu', (d.get('cpu_baseline') or {}).get('value'))
PY
}
timeout 600 python3 bench.py > $O/d1.json 2> $O/d1.err; echo "rc=$?"; show $O/d1.json default
timeout 300 python3 bench.py --cpu-baseline 0 --secondary 0 > $O/n1.json 2> $O/n1.err; show $O/n1.json no-baseline
timeout 300 python3 bench.py --cpu-baseline 0 --secondary 0 --steps 20 --warmup 5 > $O/n2.json 2> $O/n2.err; show $O/n2.json steps20
timeout 300 python3 bench.py --cpu-baseline 0 --secondary 0 --frames-in-flight 1 > $O/s1.json 2> $O/s1.err; show $O/s1.json depth1
timeout 600 python3 bench.py --steps 20 --warmup 5 > $O/d2.json 2> $O/d2.err; echo "rc=$?"; show $O/d2.json default-20-5
