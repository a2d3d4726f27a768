#!/bin/bash
O=gpurun_out/r03_pyr; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_coords.py tests/test_gpu_codec_v2.py tests/test_gpu_me_api.py tests/test_gpu_codec_color.py -x -q -m gpu 2>&1 | tail -3 | tee $O/tests.txt
timeout 200 python tools/timeline.py 2>&1 | grep -v amdgpu | tail -6 | tee $O/timeline.txt
timeout 300 python bench.py --secondary 0 --cpu-baseline 0 --steps 20 --warmup 5 2>&1 | tail -1 | tee $O/bench.json
