#!/bin/bash
# does a larger offset-split threshold (summation order 2 on bigger maps) pay?  ms/step of the headline step per threshold
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r02w; mkdir -p $O
for t in 8192 20000 80000; do
  FPCC_SPLIT_MAX_ROWS=$t timeout 200 python bench.py --steps 10 --warmup 3 --cpu-baseline 0 --secondary 0 > $O/bench_split_$t.json 2> $O/bench_split_$t.err
  python - <<PY
import json
d=json.load(open("$O/bench_split_$t.json"))
print($t, d["ms_per_step"], d["config"]["encode_ms"], d["config"]["decode_ms"], d["roofline"]["kernel_ms_per_step"], d["roofline"]["frac"], d["config"]["bytes"])
PY
done
