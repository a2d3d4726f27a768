#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r02pw; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_conv.py -x -q -k "pointwise or wave" > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt
tail -15 $O/pytest.txt
for v in 0 49152 20000 100000; do
  FPCC_POINTWISE_MIN_ROWS=$v timeout 200 python bench.py --steps 10 --warmup 3 --cpu-baseline 0 --secondary 0 > $O/bench_$v.json 2> $O/bench_$v.err
  python - <<PY
import json
d=json.load(open("$O/bench_$v.json"))
print($v, d["ms_per_step"], d["config"]["encode_ms"], d["config"]["decode_ms"], d["roofline"]["kernel_ms_per_step"], d["roofline"]["frac"], d["config"]["bytes"])
PY
done
