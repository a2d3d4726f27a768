#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
for v in 2048 1; do echo "FPCC_I8_LINEAR_TILED_MIN=$v"; FPCC_I8_LINEAR_TILED_MIN=$v timeout 300 python3 tools/r05/int_phases.py sync 2>&1 | grep -v amdgpu | cut -c1-200; done
timeout 900 python3 -m pytest tests/test_gpu_codec_int.py tests/test_gpu_int_ops.py -x -q 2>&1 | tail -3
