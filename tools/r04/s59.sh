#!/bin/bash
# GPU session 59: the driver's round-end sequence on the final commit: GPU suite, smoke, default bench
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r04Z; mkdir -p $O
timeout 1500 python3 -m pytest tests/ -x -q -m gpu > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
timeout 120 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; python3 -c "
import json; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['cpu_baseline']['value'], {k:(v.get('encode_ms'),v.get('decode_ms'),v.get('ms_per_step')) for k,v in d['config']['secondary'].items()})"
