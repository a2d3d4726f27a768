#!/bin/bash
# GPU session 51: libfpcc_hip calls with the interpreter lock held (PyDLL) against released around every launch (CDLL), alternating repeats
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r04I; mkdir -p $O
timeout 300 python3 -m pytest tests/test_gpu_serving.py tests/test_gpu_conv.py -q -x 2>&1 | tail -2
for rep in 1 2 3 4; do
for rel in 0 1; do
  FPCC_HIP_RELEASE_GIL=$rel timeout 300 python3 bench.py --cpu-baseline 0 --secondary 0 --steps 20 --warmup 5 > $O/g.json 2> $O/g.err
  python3 - <<PY
import json
d=json.loads(open('$O/g.json').read().strip().splitlines()[-1]); print('release_gil=$rel', d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_step'], d['config']['encode_ms'], d['config']['decode_ms'])
PY
done; done
FPCC_HIP_RELEASE_GIL=0 timeout 300 python3 bench.py --cpu-baseline 0 --secondary 0 --frames-in-flight 1 > $O/g.json 2>/dev/null; python3 -c "
import json; d=json.loads(open('$O/g.json').read().strip().splitlines()[-1]); print('depth1 keep', d['value'])"
FPCC_HIP_RELEASE_GIL=1 timeout 300 python3 bench.py --cpu-baseline 0 --secondary 0 --frames-in-flight 1 > $O/g.json 2>/dev/null; python3 -c "
import json; d=json.loads(open('$O/g.json').read().strip().splitlines()[-1]); print('depth1 release', d['value'])"
