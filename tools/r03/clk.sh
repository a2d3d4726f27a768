#!/bin/bash
O=gpurun_out/r03_clk; mkdir -p $O
timeout 200 python -m pytest tests/test_gpu_metrics.py -x -q -m gpu 2>&1 | tail -2
timeout 300 python bench.py --secondary 0 --cpu-baseline 0 2>&1 | tail -1 | tee $O/bench.json
timeout 300 python bench.py --secondary 0 --cpu-baseline 0 --steps 30 2>&1 | tail -1 | tee $O/bench30.json
