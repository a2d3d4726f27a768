#!/bin/bash
# result-neutral kernel knobs at the bench's batch size (16 frames): one box, the default first and last
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
run() { # name, env assignments...
  local name=$1; shift
  env "$@" timeout 600 python3 bench.py --cpu-baseline 0 --secondary 0 --latency-frames 2 --steps 8 > $O/g41_$name.json 2> $O/g41_$name.err
  python3 - $O/g41_$name.json "$name" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(f"{sys.argv[2]:28s} value {d['value']:7.2f}  frac {d['roofline']['frac']:.4f}  family ms/step {d['roofline']['kernel_ms_per_step']:.1f}")
except Exception as e:
    print(sys.argv[2], 'failed', e)
PY
}
run default_a FPCC_X=0
run sb0 FPCC_WAVE_SB=0
run sb2 FPCC_WAVE_SB=2
run persist1 FPCC_CONV_PERSIST=1
run persist2 FPCC_CONV_PERSIST=2
run fold50k FPCC_GROUPED_FOLD_ROWS=51200
run fold400k FPCC_GROUPED_FOLD_ROWS=409600
run wave22 FPCC_WAVE22_MIN_ROWS=1000000
run nbw2 FPCC_WAVE_NBW=2
run nbw4 FPCC_WAVE_NBW=4
run pw8k FPCC_POINTWISE_MIN_ROWS=8192
run default_b FPCC_X=0
