#!/bin/bash
# same-box A/B: occupancy predictors deferred (default) against chain order
O=gpurun_out/r03_defer; mkdir -p $O; rm -f $O/ab.txt
for rep in 1 2 3; do
  for d in 1 0; do
    FPCC_DEFER_OCCUPANCY=$d timeout 300 python bench.py --secondary 0 --cpu-baseline 0 --steps 20 2>&1 | tail -1 > $O/b.json
    echo "defer=$d $(grep -o '"ms_per_step": [0-9.]*\|"encode_ms": [0-9.]*\|"decode_ms": [0-9.]*\|"kernel_ms_per_step": [0-9.]*\|"shader_clock_mhz": [0-9]*' $O/b.json | tr '\n' ' ')" | tee -a $O/ab.txt
  done
done
