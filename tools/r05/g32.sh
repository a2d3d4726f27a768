#!/bin/bash
# bench.py with and without the process bound to the GPU's NUMA node (FPCC_NUMA_BIND), twice each
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
for rep in 1 2; do for b in 0 1; do
  FPCC_NUMA_BIND=$b timeout 600 python3 bench.py --cpu-baseline 0 > $O/g32_bind${b}_$rep.json 2> $O/g32_bind${b}_$rep.err
  python3 - $O/g32_bind${b}_$rep.json $b <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
s = d['config'].get('secondary', {})
print('bind', sys.argv[2], 'value', d['value'], 'one_frame', d.get('value_one_frame'), 'frac', d['roofline']['frac'],
      {k: (v.get('encode_ms'), v.get('decode_ms'), v.get('ms_per_step')) for k, v in s.items() if isinstance(v, dict)})
PY
done; done
