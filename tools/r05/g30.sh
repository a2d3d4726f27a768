#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_codec_int.py tests/test_gpu_int_ops.py tests/test_gpu_ptq.py tests/test_gpu_lossl_float.py -x -q 2>&1 | tail -3
timeout 300 python3 tools/r05/int_phases.py sync 2>&1 | grep -v amdgpu | cut -c1-200
timeout 300 python3 tools/timeline_int.py 2>&1 | tail -3
