#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
timeout 1200 python3 -m pytest tests/test_gpu_int_ops.py tests/test_gpu_codec_int.py tests/test_gpu_lossl_float.py tests/test_gpu_ptq.py -x -q 2>&1 | tail -5
timeout 300 python3 tools/r05/i8_stamp_probe.py 0,4 > $O/g11_i8_stamps.txt 2>&1; grep -E "^##|^\|" $O/g11_i8_stamps.txt
echo "== i8 probe"; timeout 200 python3 tools/r05/i8_probe.py 20 2>&1 | grep "^|" | cut -d'|' -f2,3,6,7 | tee $O/g11_i8.txt
TOP=12 python3 tools/int_launches.py 2>&1 | grep -E "^==|k_conv_i8_tiled"
