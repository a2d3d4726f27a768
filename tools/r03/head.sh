#!/bin/bash
O=gpurun_out/r03_head; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_mlp_chain.py tests/test_gpu_codec_v2.py tests/test_gpu_fullsize.py -x -q -m gpu > $O/pytest.txt 2>&1; echo "pytest rc $?" >> $O/pytest.txt
tail -5 $O/pytest.txt
for i in 1 2; do timeout 300 python bench.py --steps 20 --warmup 5 --secondary 0 --cpu-baseline 0 > $O/bench$i.json 2> $O/bench$i.err; python - <<PY
import json
d=json.load(open('$O/bench$i.json'))
print('bench', d['value'], d['ms_per_step'], d['config']['encode_ms'], d['config']['decode_ms'], d['roofline']['frac'], d['roofline']['kernel_ms_per_step'], d['roofline']['other_conv_ms_per_step'])
PY
done
