#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r02h; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_conv.py -x -q 2>&1 | tail -3
run() { timeout 300 python tools/conv_probe.py "$@" 2>&1 | grep -v amdgpu.ids; }
ONLY=pattern NBWS=0,2,4 run 1 128 128 20 | tee $O/l1.txt
ONLY=pattern NBWS=0,1,2 run 2 128 128 20 | tee $O/l2.txt
ONLY=pattern NBWS=0,1,2 run 3 128 128 20 | tee $O/l3.txt
ONLY=pattern NBWS=0,1,2 run 1 64 64 20 | tee $O/l1_64.txt
timeout 300 python bench.py --steps 20 --warmup 5 --cpu-baseline 0 --secondary 0 --dump-trace $O/conv_launches.txt > $O/bench.json 2> $O/bench.err
python -c "
import json
d = json.loads(open('$O/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['config']['encode_ms'], d['config']['decode_ms'], d['roofline']['frac'], d['roofline']['kernel_ms_per_step'])"
