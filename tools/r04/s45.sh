#!/bin/bash
# GPU session 45: row-order window 2^17 against 2^19 on the colour codec (544 K- / 1.1 M- / 4.36 M-row maps) and the 2 M-voxel v2 frame
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
for w in 17 19 17 19; do
  echo "== window 2^$w"; FPCC_ROW_WINDOW_LOG2=$w timeout 300 python3 tools/timeline_color.py 2>/dev/null | tail -3 | head -2
done
show() { python3 - <<PY
import json
d=json.loads(open('$1').read().strip().splitlines()[-1]); r=d['roofline']
print('$2', d['value'], d['ms_per_step'], 'frac', r['frac'], 'kernel_ms', r['kernel_ms_per_step'])
PY
}
O=gpurun_out/r04I; mkdir -p $O
for w in 17 19; do
  FPCC_ROW_WINDOW_LOG2=$w timeout 300 python3 bench.py --cpu-baseline 0 --secondary 0 --steps 10 --warmup 3 --resolution 2048 > $O/r$w.json 2> $O/r$w.err; show $O/r$w.json "2048^3 frame, window 2^$w"
done
