#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r03_train1; mkdir -p $O
for i in 1 2; do timeout 300 python bench.py --steps 20 --warmup 5 --secondary 0 --cpu-baseline 0 > $O/bench$i.json 2> $O/bench$i.err; python - <<PY
import json
d=json.load(open('$O/bench$i.json'))
print('bench', d['value'], d['ms_per_step'], d['config']['encode_ms'], d['config']['decode_ms'], d['roofline']['frac'], d['roofline']['kernel_ms_per_step'])
PY
done
timeout 400 rocprofv3 --kernel-trace --stats -d $O/train -o s --output-format csv -- python3 bench_train.py --steps 4 --warmup 1 > $O/train.log 2>&1
tail -1 $O/train.log | head -c 400; echo
python profiles/summarize.py $(find $O/train -name '*kernel_stats.csv' | head -1) 5 | head -36
