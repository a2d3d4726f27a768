#!/bin/bash
O=gpurun_out/r03_chain2; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_mlp_chain.py -x -q -m gpu > $O/pytest.txt 2>&1; echo "pytest rc $?" >> $O/pytest.txt
tail -5 $O/pytest.txt
timeout 300 python tools/mlp_chain_probe.py 30 2>&1 | grep -v amdgpu.ids | tee $O/probe.txt
timeout 600 python bench.py --steps 20 --warmup 5 --secondary 0 --cpu-baseline 0 --dump-trace $O/conv_launches.txt > $O/bench.json 2> $O/bench.err; echo "bench rc $?" >> $O/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03_chain2/bench.json'))
print(d['value'], d['ms_per_step'], d['config']['encode_ms'], d['config']['decode_ms'], d['roofline']['frac'], d['roofline']['kernel_ms_per_step'], d['roofline']['launches_per_step'], d['roofline']['algorithmic_gflop_per_step'])
PY
