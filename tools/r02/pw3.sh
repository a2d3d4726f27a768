#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r02pw; mkdir -p $O
for v in 0 32768 16384 0 32768; do
  FPCC_POINTWISE_MIN_ROWS=$v timeout 200 python bench.py --steps 10 --warmup 3 --cpu-baseline 0 --secondary 0 > $O/bench3_$v.json 2> $O/bench3_$v.err
  python - <<PY
import json
d=json.load(open("$O/bench3_$v.json"))
print($v, d["ms_per_step"], d["config"]["encode_ms"], d["config"]["decode_ms"], d["roofline"]["kernel_ms_per_step"], d["roofline"]["frac"], d["config"]["bytes"])
PY
done
