#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
for v in "4 2" "2 4" "2 3"; do set -- $v
  echo "== i8 probe NB=$1 MW=$2"; FPCC_I8_NB=$1 FPCC_I8_MW=$2 timeout 200 python3 tools/r05/i8_probe.py 20 2>&1 | grep "^|" | cut -d'|' -f2,3,6,7 | tee $O/g8_i8_nb$1_mw$2.txt
done
FPCC_I8_NB=2 timeout 600 python3 -m pytest tests/test_gpu_int_ops.py tests/test_gpu_codec_int.py -x -q 2>&1 | tail -3
python3 bench.py --steps 10 --warmup 3 --batch 8 --frames-in-flight 2 --own-streams 1 --secondary 0 --cpu-baseline 0 > $O/g8_bench_b8_o1.json 2> $O/g8_bench_b8_o1.err
python3 -c "
import json
d = json.loads(open('$O/g8_bench_b8_o1.json').read().strip().splitlines()[-1])
print('batch 8 depth 2 own-streams 1: value', d['value'], 'one_frame', d['value_one_frame'], 'ms/step', d['ms_per_step'], 'frac', d['roofline']['frac'])"
