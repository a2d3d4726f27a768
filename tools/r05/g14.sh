#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_codec_many.py tests/test_gpu_codec_v2.py tests/test_gpu_serving.py -x -q 2>&1 | tail -4
timeout 600 python3 bench.py --secondary 0 --cpu-baseline 0 > $O/g14_bench.json 2> $O/g14_bench.err; tail -3 $O/g14_bench.err
python3 -c "
import json
d = json.loads(open('$O/g14_bench.json').read().strip().splitlines()[-1])
r = d['roofline']
print('value', d['value'], 'one_frame', d['value_one_frame'], 'ms/step', d['ms_per_step'], 'frac', r['frac'], 'clock', r['shader_clock_mhz'], 'frac@clock', r['frac_at_shader_clock'], 'traffic', r['traffic'], 'alg', r['algorithmic_bytes_per_launch'], r['traffic_source'])"
