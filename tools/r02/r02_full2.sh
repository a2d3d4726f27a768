#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r02f; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_fullsize.py -x -q --durations=6 > $O/pytest_fullsize.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_fullsize.txt
tail -25 $O/pytest_fullsize.txt
( time python bench.py ) > $O/bench_default.json 2> $O/bench_default.err
tail -4 $O/bench_default.err
python -c "
import json
d = json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'])
print(json.dumps(d['config']['secondary'], indent=1))
print(json.dumps(d['cpu_baseline'], indent=1))"
