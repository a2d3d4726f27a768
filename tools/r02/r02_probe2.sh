#!/bin/bash
# wave kernel: correctness (conv tests) then a sweep of unit width / scheduling per pyramid level, then the whole bench
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r02b; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_conv.py -x -q > $O/pytest_conv.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_conv.txt
tail -5 $O/pytest_conv.txt
for lvl in 1 2 3 4 5; do
  SWEEP=1 ONLY=pattern timeout 300 python tools/conv_probe.py $lvl 128 128 20 > $O/sweep_l$lvl.txt 2>&1
done
SWEEP=1 ONLY=pattern timeout 300 python tools/conv_probe.py 2 256 128 20 > $O/sweep_l2_256.txt 2>&1
SWEEP=1 ONLY=pattern timeout 300 python tools/conv_probe.py 1 64 64 20 > $O/sweep_l1_64.txt 2>&1
cat $O/sweep_*.txt | grep -v amdgpu.ids
FPCC_CONV_WAVE=0 python bench.py --steps 10 --warmup 3 --cpu-baseline 0 > $O/bench_tiled.json 2> $O/bench_tiled.err
python bench.py --steps 10 --warmup 3 --cpu-baseline 0 --dump-trace $O/conv_launches_wave.txt > $O/bench_wave.json 2> $O/bench_wave.err
cat $O/bench_tiled.json $O/bench_wave.json | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['value'], d['ms_per_step'], d['config']['encode_ms'], d['config']['decode_ms'], d['roofline']['frac'], d['roofline']['kernel_ms_per_step'], d['config']['bytes'])"
tail -3 $O/bench_wave.err
