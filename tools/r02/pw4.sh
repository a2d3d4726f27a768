#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r02pw; mkdir -p $O
for v in 0 32768 12288; do
  FPCC_POINTWISE_MIN_ROWS=$v timeout 200 python bench.py --steps 6 --warmup 2 --cpu-baseline 0 --secondary 0 --dump-trace $O/trace_$v.txt > $O/bench4_$v.json 2> $O/bench4_$v.err
  python - <<PY
rows=[l.split() for l in open("$O/trace_$v.txt")][1:]
import collections
acc=collections.defaultdict(float)
for r in rows:
    if r[0]!='mfma': continue
    n=int(r[3]); b='>=200K' if n>=200000 else '50-200K' if n>=50000 else '12-50K' if n>=12000 else '<12K'
    acc[(b, r[4])]+=float(r[6])
print($v, 'pointwise ms by size:', {k[0]: round(v,3) for k,v in sorted(acc.items()) if k[1]=='1'}, 'all mfma', round(sum(acc.values()),3))
PY
done
