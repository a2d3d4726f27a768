#!/bin/bash
# GPU session 31: bench with the whole timed region pipelined (traced steps inside the pipeline)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r04G; mkdir -p $O
for cfg in "10 3" "20 5" "10 3"; do
  set -- $cfg
  timeout 300 python3 bench.py --steps $1 --warmup $2 --cpu-baseline 0 --secondary 0 > $O/b_$1.json 2> $O/b_$1.err; echo "rc=$?"
  python3 - <<PY
import json
d=json.loads(open('$O/b_$1.json').read().strip().splitlines()[-1]); r=d['roofline']; c=d['config']
print('steps $1', d['value'], d['ms_per_step'], 'enc', c['encode_ms'], 'dec', c['decode_ms'], 'frac', r['frac'], 'kernel_ms', r['kernel_ms_per_step'], 'launches', r['launches_per_step'], 'clock', r['shader_clock_mhz'])
PY
done
timeout 300 python3 bench.py --steps 10 --warmup 3 --cpu-baseline 0 --secondary 0 --frames-in-flight 1 > $O/b1.json 2> $O/b1.err; head -c 200 $O/b1.json; echo
timeout 300 python3 -m pytest tests/test_gpu_bench_ddp.py -q 2>&1 | tail -2
