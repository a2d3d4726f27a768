#!/bin/bash
# GPU session 2 (round 4): k_conv_lds -- parity, then timing against the default kernels
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r04b; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_conv.py -x -q -k "lds_operand or folded" > $O/pytest.txt 2>&1
tail -5 $O/pytest.txt
for shape in "1 128 128" "2 128 128" "1 64 64" "2 256 128" "1 128 64" "1 64 128" "3 128 128"; do
  ONLY=pattern LDS=0,2,3,4 timeout 300 python3 tools/conv_probe.py $shape 20 >> $O/lds.txt 2>&1
done
ONLY=pattern LDS=2,3,4 DBG=1,2,3 timeout 300 python3 tools/conv_probe.py 1 128 128 20 >> $O/lds_dbg.txt 2>&1
cat $O/lds.txt
