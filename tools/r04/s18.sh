#!/bin/bash
# GPU session 18: colour decoder vs chain-order oracle diagnosis; remaining suites after numerics version 3
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r04t; mkdir -p $O
timeout 600 python3 tools/r04/col_diag.py > $O/col_diag.txt 2>&1; tail -15 $O/col_diag.txt
timeout 900 python3 -m pytest tests/test_gpu_codec_color.py -q > $O/color.txt 2>&1; tail -6 $O/color.txt
timeout 900 python3 -m pytest tests/test_gpu_entropy_glue.py tests/test_gpu_rans_dev.py -x -q > $O/ent.txt 2>&1; tail -6 $O/ent.txt
