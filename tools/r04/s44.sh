#!/bin/bash
# GPU session 44: row-order window 2^17 (default) against 2^18 / 2^20 (one window = the whole 272 K-row map): the layer alone and the bench
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r04I; mkdir -p $O
for w in 17 18 20; do
  echo "== FPCC_ROW_WINDOW_LOG2=$w"
  FPCC_ROW_WINDOW_LOG2=$w ONLY=pattern timeout 200 python3 tools/conv_probe.py 1 128 128 20 2>/dev/null | grep level
  FPCC_ROW_WINDOW_LOG2=$w ONLY=pattern timeout 200 python3 tools/conv_probe.py 1 256 128 20 2>/dev/null | grep level
done
show() { python3 - <<PY
import json
d=json.loads(open('$1').read().strip().splitlines()[-1]); r=d['roofline']
print('$2', d['value'], d['ms_per_step'], 'frac', r['frac'], 'kernel_ms', r['kernel_ms_per_step'])
PY
}
for w in 17 20 17 20; do
  FPCC_ROW_WINDOW_LOG2=$w timeout 300 python3 bench.py --cpu-baseline 0 --secondary 0 --steps 20 --warmup 5 > $O/w$w.json 2> $O/w$w.err; show $O/w$w.json "window 2^$w"
done
