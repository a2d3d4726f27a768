#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
echo "== i8 probe, full"; timeout 200 python3 tools/r05/i8_probe.py 20 2>&1 | grep "^|" | tee $O/g7_i8_full.txt
echo "== i8 probe, no epilogue (FPCC_I8_DBG=8)"; FPCC_I8_DBG=8 timeout 200 python3 tools/r05/i8_probe.py 20 2>&1 | grep "^|" | tee $O/g7_i8_noepi.txt
echo "== i8 probe, no MFMA/gather/W (FPCC_I8_DBG=7)"; FPCC_I8_DBG=7 timeout 200 python3 tools/r05/i8_probe.py 20 2>&1 | grep "^|" | tee $O/g7_i8_dbg7.txt
echo "== wgrad by shape"; timeout 300 python3 tools/train_wgrad_trace.py > $O/g7_wgrad.txt 2>&1; tail -30 $O/g7_wgrad.txt
for cfg in "4 2 1" "8 2 0"; do set -- $cfg
  python3 bench.py --steps 12 --warmup 3 --batch $1 --frames-in-flight $2 --own-streams $3 --secondary 0 --cpu-baseline 0 > $O/g7_bench_b$1_d$2_o$3.json 2> $O/g7_bench_b$1_d$2_o$3.err
  python3 -c "
import json
d = json.loads(open('$O/g7_bench_b$1_d$2_o$3.json').read().strip().splitlines()[-1])
print('batch $1 depth $2 own-streams $3: value', d['value'], 'one_frame', d['value_one_frame'], 'ms/step', d['ms_per_step'], 'frac', d['roofline']['frac'])"
done
