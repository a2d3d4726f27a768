#!/bin/bash
# GPU session 37: contexts share parameter OBJECTS (no re-packing when they alternate): serving tests, bench, pack launches per step
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r04H; mkdir -p $O
timeout 600 python3 -m pytest tests/test_gpu_serving.py -q 2>&1 | tail -2
show() { python3 - <<PY
import json
d=json.loads(open('$1').read().strip().splitlines()[-1]); r=d['roofline']; c=d['config']
print('$2', d['value'], d['ms_per_step'], 'frac', r['frac'], 'kernel_ms', r['kernel_ms_per_step'])
PY
}
for i in 1 2; do timeout 600 python3 bench.py --cpu-baseline 0 --secondary 0 > $O/p$i.json 2> $O/p$i.err; show $O/p$i.json "neither"; done
timeout 600 python3 bench.py --cpu-baseline 0 --secondary 0 --steps 20 --warmup 5 > $O/p3.json 2> $O/p3.err; show $O/p3.json "20 5"
timeout 300 rocprofv3 --kernel-trace --stats -d $O/st -o s --output-format csv -- python3 bench.py --steps 10 --warmup 3 --cpu-baseline 0 --secondary 0 > $O/st.log 2>&1
python3 profiles/summarize.py $(find $O/st -name s_kernel_stats.csv | head -1) 17 | grep -i "pack_weights\|all kernels\|copyBuffer\|vectorized_gather\|transpose_table"
find $O/st -name 's_kernel_trace.csv' -delete
