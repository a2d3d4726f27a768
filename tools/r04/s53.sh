#!/bin/bash
# GPU session 53: distribution of the default bench line at the final code state (driver's flags 20 / 5 and the defaults), no secondary legs
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r04I; mkdir -p $O
for i in 1 2 3 4 5 6; do
  timeout 300 python3 bench.py --cpu-baseline 0 --secondary 0 --steps 20 --warmup 5 > $O/z.json 2>/dev/null; python3 -c "
import json; d=json.loads(open('$O/z.json').read().strip().splitlines()[-1]); print('20/5', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_ms_per_step'])"
  timeout 300 python3 bench.py --cpu-baseline 0 --secondary 0 > $O/z.json 2>/dev/null; python3 -c "
import json; d=json.loads(open('$O/z.json').read().strip().splitlines()[-1]); print('10/3', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_ms_per_step'])"
done
