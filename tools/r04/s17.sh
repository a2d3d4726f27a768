#!/bin/bash
# GPU session 17: numerics version 3 (specified logistic function): new GPU golden stream, strict chain-order parity, v2 / colour suites
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r04s; mkdir -p $O
python3 tools/make_gpu_golden.py $O/v2_stream.json > $O/golden.txt 2>&1; tail -2 $O/golden.txt
cp $O/v2_stream.json tests/golden/v2_stream.json
timeout 900 python3 -m pytest tests/test_gpu_codec_v2.py -x -q > $O/v2.txt 2>&1; tail -12 $O/v2.txt
timeout 900 python3 -m pytest tests/test_gpu_fullsize.py -x -q -k "committed or earlier or oracle" > $O/full.txt 2>&1; tail -6 $O/full.txt
timeout 900 python3 -m pytest tests/test_gpu_codec_color.py tests/test_gpu_entropy_glue.py tests/test_gpu_rans_dev.py -x -q > $O/color.txt 2>&1; tail -6 $O/color.txt
