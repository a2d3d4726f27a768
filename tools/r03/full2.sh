#!/bin/bash
O=gpurun_out/r03_full2; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu -x > $O/pytest.txt 2>&1; echo "pytest rc $?" >> $O/pytest.txt
tail -5 $O/pytest.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?" >> $O/smoke.txt; tail -2 $O/smoke.txt
timeout 600 python bench.py --steps 20 --warmup 5 --dump-trace $O/conv_launches.txt > $O/bench.json 2> $O/bench.err; echo "bench rc $?" >> $O/bench.err
head -c 2600 $O/bench.json
