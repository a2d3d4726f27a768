#!/bin/bash
# GPU session 10: frames in flight (threads, one stream): bench with depth 1 / 2 / 3; a quick parity check of the pipelined contexts
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r04l; mkdir -p $O
for d in 1 2; do
  timeout 600 python3 bench.py --steps 20 --warmup 5 --cpu-baseline 0 --secondary 0 --frames-in-flight $d > $O/d$d.json 2> $O/d$d.err
  [ $d = 2 ] && timeout 600 python3 bench.py --steps 20 --warmup 5 --cpu-baseline 0 --secondary 0 --frames-in-flight 2 --own-streams 1 > $O/d3.json 2> $O/d3.err
done
python3 - <<'PY'
import json
for n in ('d1', 'd2', 'd3'):
    try:
        d = json.loads(open(f'gpurun_out/r04l/{n}.json').read().strip().splitlines()[-1])
        print(n, d['value'], 'Mpoints/s', d['ms_per_step'], 'ms/step enc', d['config']['encode_ms'], 'dec', d['config']['decode_ms'], 'frac', d['roofline']['frac'], 'kernel ms', d['roofline']['kernel_ms_per_step'], 'clock', d['roofline']['shader_clock_mhz'])
    except Exception as e:
        print(n, 'failed', e); print(open(f'gpurun_out/r04l/{n}.err').read()[-1500:])
PY
