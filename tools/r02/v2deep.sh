#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r02d; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_codec_v2.py tests/test_gpu_me_api.py tests/test_gpu_codec_color.py -x -q --durations=5 > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt
tail -40 $O/pytest.txt
