#!/bin/bash
# GPU session 25: kernel durations (not wall clock) of the int8 tiled convolution per level, ablations
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r04z; mkdir -p $O
for d in 0 7 1; do
  export FPCC_I8_DBG=$d
  timeout 300 rocprofv3 --kernel-trace -d $O/tr$d -o p --output-format csv -- python3 tools/r04/i8_probe.py 8 > $O/tr$d.log 2>&1
  echo "== FPCC_I8_DBG=$d"; python3 tools/r04/i8_trace_parse.py $(find $O/tr$d -name 'p_kernel_trace.csv' | head -1)
done
