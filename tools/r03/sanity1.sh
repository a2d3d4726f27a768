#!/bin/bash
# round 3, first GPU call: changed tests + the baseline bench of the round's starting state
O=gpurun_out/r03_sanity1; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_me_api.py tests/test_gpu_rans_dev.py tests/test_gpu_metrics.py tests/test_gpu_conv.py -x -q -m gpu > $O/pytest.txt 2>&1; echo "pytest rc $?" >> $O/pytest.txt
timeout 600 python bench.py --steps 20 --warmup 5 --dump-trace $O/conv_launches.txt > $O/bench.json 2> $O/bench.err; echo "bench rc $?" >> $O/bench.err
timeout 300 python bench_train.py --steps 10 --warmup 3 > $O/train.json 2> $O/train.err
tail -3 $O/pytest.txt; cat $O/bench.json | head -c 1500; echo; cat $O/train.json | head -c 600
