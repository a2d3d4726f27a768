#!/bin/bash
# full GPU suite + smoke + the default bench line of the committed state
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r02z; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q --durations=8 > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt
tail -14 $O/pytest_gpu.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?" >> $O/smoke.txt; tail -2 $O/smoke.txt
timeout 480 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; cat $O/bench_default.json
