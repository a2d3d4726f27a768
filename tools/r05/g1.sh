#!/bin/bash
# first GPU session of round 5: the batched traversal's parity tests, the v2 tests it touches, and its throughput against B
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r05
python -m pytest tests/test_gpu_codec_many.py tests/test_engine_managers.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r05/g1_many.log
python -m pytest tests/test_gpu_codec_v2.py tests/test_gpu_coords.py tests/test_gpu_codec_color.py tests/test_gpu_serving.py tests/test_gpu_training.py -x -q -m gpu 2>&1 | tail -8 > gpurun_out/r05/g1_v2.log
python tools/r05/many_bench.py 8 1,2,3,4 > gpurun_out/r05/g1_bench.log 2>&1
tail -5 gpurun_out/r05/g1_many.log gpurun_out/r05/g1_v2.log; cat gpurun_out/r05/g1_bench.log
