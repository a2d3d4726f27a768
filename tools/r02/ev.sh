#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r02ev; mkdir -p $O
for i in 1 2 3 4; do
  timeout 200 python bench.py --steps 10 --warmup 3 --cpu-baseline 0 --secondary 0 --dump-trace $O/trace_$i.txt > $O/bench_$i.json 2> $O/bench_$i.err
  python - <<PY
import json
d=json.load(open("$O/bench_$i.json"))
slow=[l.split() for l in open("$O/trace_$i.txt") if l.startswith('mfma 256 128 70530')]
print($i, d["ms_per_step"], d["roofline"]["kernel_ms_per_step"], d["roofline"]["frac"], [r[6] for r in slow])
PY
done
