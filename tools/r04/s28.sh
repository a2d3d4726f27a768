#!/bin/bash
# GPU session 28: int8 tiled convolution, final form of the round: per-level table (kernel durations), integer codec timeline, all int tests
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r04z; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_int_ops.py tests/test_gpu_codec_int.py tests/test_gpu_lossl_float.py -x -q > $O/int_tests.txt 2>&1; tail -3 $O/int_tests.txt
timeout 300 rocprofv3 --kernel-trace -d $O/trf -o p --output-format csv -- python3 tools/r04/i8_probe.py 8 > $O/trf.log 2>&1
grep "^|" $O/trf.log > $O/i8_probe_table.txt
python3 tools/r04/i8_trace_parse.py $(find $O/trf -name 'p_kernel_trace.csv' | head -1) | tee $O/i8_durations.txt
timeout 300 python3 tools/timeline_int.py > $O/int_timeline3.txt 2>&1; tail -2 $O/int_timeline3.txt
