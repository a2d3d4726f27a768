#!/bin/bash
# full GPU suite of the round's state: golden stream of numerics version 2 regenerated first, then pytest -m gpu, then the bench
O=gpurun_out/r03_full1; mkdir -p $O
python tools/make_gpu_golden.py $O/v2_stream.json > $O/golden.txt 2>&1 && cp $O/v2_stream.json tests/golden/v2_stream.json
tail -2 $O/golden.txt
timeout 2400 python -m pytest tests -q -m gpu -x > $O/pytest.txt 2>&1; echo "pytest rc $?" >> $O/pytest.txt
tail -8 $O/pytest.txt
timeout 600 python bench.py --steps 20 --warmup 5 --dump-trace $O/conv_launches.txt > $O/bench.json 2> $O/bench.err; echo "bench rc $?" >> $O/bench.err
head -c 1800 $O/bench.json
