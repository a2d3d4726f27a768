#!/bin/bash
# GPU session 47: interpreter switch interval of the frame pipeline (default 2e-4 s)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r04I; mkdir -p $O
show() { python3 - <<PY
import json
d=json.loads(open('$1').read().strip().splitlines()[-1]); r=d['roofline']
print('$2', d['value'], d['ms_per_step'], 'frac', r['frac'], 'kernel_ms', r['kernel_ms_per_step'])
PY
}
for sw in 2e-4 5e-5 1e-3 2e-5 2e-4 5e-5; do
  FPCC_SWITCH_INTERVAL=$sw timeout 300 python3 bench.py --cpu-baseline 0 --secondary 0 --steps 20 --warmup 5 > $O/sw.json 2> $O/sw.err; show $O/sw.json "switch interval $sw"
done
