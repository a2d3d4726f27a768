#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r02e; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt
tail -6 $O/pytest_gpu.txt
python bench.py --steps 20 --warmup 5 --cpu-baseline 0 --dump-trace $O/conv_launches.txt > $O/bench.json 2> $O/bench.err
FPCC_CONV_WAVE=0 python bench.py --steps 20 --warmup 5 --cpu-baseline 0 > $O/bench_tiled.json 2> $O/bench_tiled.err
cat $O/bench_tiled.json $O/bench.json | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['value'], d['ms_per_step'], d['config']['encode_ms'], d['config']['decode_ms'], d['roofline']['frac'], d['roofline']['kernel_ms_per_step'], d['config']['bytes'])"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
