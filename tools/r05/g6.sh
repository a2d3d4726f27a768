#!/bin/bash
# full GPU suite + smoke + training bench after the round's engine changes
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
timeout 1700 python3 -m pytest tests -m gpu -x -q --durations=8 > $O/g6_pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/g6_pytest_gpu.txt
tail -15 $O/g6_pytest_gpu.txt
timeout 120 python3 -c "import __graft_entry__ as g; g.smoke()" > $O/g6_smoke.txt 2>&1; echo "smoke rc=$?" >> $O/g6_smoke.txt; tail -2 $O/g6_smoke.txt
timeout 300 python3 bench_train.py --steps 8 --warmup 3 > $O/g6_train.json 2> $O/g6_train.err; cat $O/g6_train.json | head -c 600
