#!/bin/bash
# neighbour-pattern row order: window size sweep at the bench's batch size (FPCC_ROW_WINDOW_LOG2; results are independent of it)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
for w in 19 13 15 17 21; do
  FPCC_ROW_WINDOW_LOG2=$w timeout 600 python3 bench.py --cpu-baseline 0 --secondary 0 > $O/g33_w$w.json 2> $O/g33_w$w.err
  python3 - $O/g33_w$w.json $w <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('window 2^' + sys.argv[2], 'value', d['value'], 'one_frame', d.get('value_one_frame'), 'frac', d['roofline']['frac'], 'achieved', d['roofline']['achieved'], 'enc/dec', d.get('encode_ms'), d.get('decode_ms'))
PY
done
