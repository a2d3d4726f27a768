#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r04i; mkdir -p $O
for shape in "1 128 128" "1 64 64" "1 256 128"; do
  FPCC_CONV_PERSIST=0 ONLY=pattern DBG=0,64,0,64 timeout 300 python3 tools/conv_probe.py $shape 20 >> $O/wpw.txt 2>&1
done
FPCC_GROUPED_FOLD_ROWS=1 FPCC_CONV_PERSIST=0 ONLY=pattern DBG=0,64 timeout 300 python3 tools/conv_probe.py 2 128 128 20 >> $O/wpw.txt 2>&1
FPCC_GROUPED_FOLD_ROWS=0 FPCC_CONV_PERSIST=0 ONLY=pattern DBG=0 timeout 300 python3 tools/conv_probe.py 1 128 128 20 >> $O/wpw.txt 2>&1
grep -v amdgpu $O/wpw.txt
