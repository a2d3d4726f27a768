#!/bin/bash
# GPU session 50: default bench line and the one-frame-at-a-time line at the final code state (switch interval 0.1 ms)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r04F; mkdir -p $O
timeout 600 python3 bench.py --dump-trace $O/conv_launches.txt > $O/bench_default.json 2> $O/bench_default.err; echo "rc=$?"
timeout 300 python3 bench.py --frames-in-flight 1 --cpu-baseline 0 --secondary 0 > $O/bench_depth1.json 2> $O/bench_depth1.err
for i in 1 2 3; do timeout 300 python3 bench.py --cpu-baseline 0 --secondary 0 > $O/x.json 2>/dev/null; python3 -c "
import json; d=json.loads(open('$O/x.json').read().strip().splitlines()[-1]); print('repeat', d['value'], d['ms_per_step'], d['roofline']['frac'])"; done
python3 -c "
import json
for n in ('bench_default','bench_depth1'):
    d=json.loads(open('$O/'+n+'.json').read().strip().splitlines()[-1]); print(n, d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_ms_per_step'])"
