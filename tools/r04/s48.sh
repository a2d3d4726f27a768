#!/bin/bash
# GPU session 48: frame-size sweep of the headline bench (same generator), two frames in flight and one at a time
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r04I; mkdir -p $O
show() { python3 - <<PY
import json
d=json.loads(open('$1').read().strip().splitlines()[-1]); r=d['roofline']; c=d['config']
print('$2', c['workload'].split(',')[1].strip(), '|', d['value'], 'Mpoints/s', d['ms_per_step'], 'ms | frac', r['frac'], '| kernel_ms', r['kernel_ms_per_step'], '| enc', c['encode_ms'], 'dec', c['decode_ms'])
PY
}
for res in 512 1024 2048; do
  for d in 2 1; do
    timeout 300 python3 bench.py --cpu-baseline 0 --secondary 0 --steps 10 --warmup 3 --resolution $res --frames-in-flight $d > $O/fs.json 2> $O/fs.err; show $O/fs.json "res $res depth $d"
  done
done
