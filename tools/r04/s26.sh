#!/bin/bash
# GPU session 26: int8 tiled convolution with the lean epilogue; ablations without epilogue; int tests
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r04z; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_int_ops.py tests/test_gpu_codec_int.py -x -q > $O/int_tests.txt 2>&1; tail -3 $O/int_tests.txt
for d in 0 8 15; do
  export FPCC_I8_DBG=$d
  timeout 300 rocprofv3 --kernel-trace -d $O/trb$d -o p --output-format csv -- python3 tools/r04/i8_probe.py 8 > $O/trb$d.log 2>&1
  echo "== FPCC_I8_DBG=$d"; python3 tools/r04/i8_trace_parse.py $(find $O/trb$d -name 'p_kernel_trace.csv' | head -1)
done
