#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r02j; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_conv.py -x -q 2>&1 | tail -3
run() { timeout 300 python tools/conv_probe.py "$@" 2>&1 | grep -v amdgpu.ids; }
ONLY=pattern NBWS=0,2,4 run 1 128 128 20 | tee $O/l1.txt
ONLY=pattern NBWS=1,2 run 1 64 64 20 | tee $O/l1_64.txt
ONLY=pattern NBWS=1,2 RINGS=1,3 run 2 128 128 20 | tee $O/l2.txt
ONLY=pattern NBWS=1,2 run 2 256 128 20 | tee $O/l2_256.txt
ONLY=pattern NBWS=1,2 RINGS=1,2 run 3 128 128 20 | tee $O/l3.txt
for lvl in 4 5 6; do
  ONLY=natural NBWS=0 run $lvl 128 128 30 | tee $O/l${lvl}_split.txt
  FPCC_SPLIT_MAX_ROWS=0 ONLY=natural NBWS=1 RINGS=1,2 run $lvl 128 128 30 | tee $O/l${lvl}_wave.txt
done
timeout 300 python bench.py --steps 20 --warmup 5 --cpu-baseline 0 --secondary 0 --dump-trace $O/conv_launches.txt > $O/bench.json 2> $O/bench.err
FPCC_CONV_WAVE=0 timeout 300 python bench.py --steps 20 --warmup 5 --cpu-baseline 0 --secondary 0 > $O/bench_tiled.json 2> $O/bench_tiled.err
cat $O/bench_tiled.json $O/bench.json | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['value'], d['ms_per_step'], d['config']['encode_ms'], d['config']['decode_ms'], d['roofline']['frac'], d['roofline']['kernel_ms_per_step'], d['config']['bytes'])"
