#!/bin/bash
# GPU session 52: with the lock kept during launches: switch interval and pipeline depth again
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r04I; mkdir -p $O
run() { timeout 300 python3 bench.py --cpu-baseline 0 --secondary 0 --steps 20 --warmup 5 $2 > $O/h.json 2> $O/h.err
  python3 - <<PY
import json
d=json.loads(open('$O/h.json').read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_step'])
PY
}
for rep in 1 2 3; do
  FPCC_SWITCH_INTERVAL=1e-4 run "interval 1e-4 depth 2" ""
  FPCC_SWITCH_INTERVAL=3e-4 run "interval 3e-4 depth 2" ""
  FPCC_SWITCH_INTERVAL=5e-5 run "interval 5e-5 depth 2" ""
  FPCC_SWITCH_INTERVAL=1e-4 run "interval 1e-4 depth 3" "--frames-in-flight 3"
done
