#!/bin/bash
# GPU session 9: does a second frame in flight fill the host-bound gaps? one bench process alone, then two at once on the same GPU
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r04k; mkdir -p $O
python3 bench.py --steps 40 --warmup 5 --cpu-baseline 0 --secondary 0 > $O/one.json 2> $O/one.err
python3 bench.py --steps 40 --warmup 5 --cpu-baseline 0 --secondary 0 > $O/two_a.json 2> $O/two_a.err &
python3 bench.py --steps 40 --warmup 5 --cpu-baseline 0 --secondary 0 > $O/two_b.json 2> $O/two_b.err &
wait
python3 - <<'PY'
import json
for n in ('one', 'two_a', 'two_b'):
    try:
        d = json.loads(open(f'gpurun_out/r04k/{n}.json').read().strip().splitlines()[-1])
        print(n, d['value'], 'Mpoints/s', d['ms_per_step'], 'ms/step enc', d['config']['encode_ms'], 'dec', d['config']['decode_ms'], 'frac', d['roofline']['frac'], 'kernel ms', d['roofline']['kernel_ms_per_step'], 'clock', d['roofline']['shader_clock_mhz'])
    except Exception as e:
        print(n, 'failed', e)
PY
