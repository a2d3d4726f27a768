#!/bin/bash
# folded (one wave per unit) against four-wave evaluation of order 3; parity tests; bench
O=gpurun_out/r03_fold; mkdir -p $O; rm -f $O/probe.txt
timeout 900 python -m pytest tests/test_gpu_conv.py -x -q -m gpu 2>&1 | tail -2 | tee $O/tests.txt
for lvl in 1 2 3; do
  for shape in "128 128" "256 128" "64 64"; do
    for fold in 1 0; do
      FPCC_GROUPED_FOLD_ROWS=$fold ONLY=pattern timeout 200 python tools/conv_probe.py $lvl $shape 20 2>&1 | grep -v amdgpu.ids | sed "s/^/fold_rows=$fold /" | tee -a $O/probe.txt
    done
  done
done
timeout 300 python bench.py --secondary 0 --cpu-baseline 0 2>&1 | tail -1 | tee $O/bench.json
FPCC_GROUPED_FOLD_ROWS=0 timeout 300 python bench.py --secondary 0 --cpu-baseline 0 2>&1 | tail -1 | tee $O/bench_nofold.json
FPCC_GROUPED_FOLD_ROWS=50000 timeout 300 python bench.py --secondary 0 --cpu-baseline 0 2>&1 | tail -1 | tee $O/bench_fold50k.json
