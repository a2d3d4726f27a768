#!/bin/bash
# GPU session 6 (round 4): persistent grouped / folded kernels with prefetched table rows: parity, then timing
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r04g; mkdir -p $O
timeout 1500 python3 -m pytest tests/test_gpu_conv.py -x -q > $O/pytest.txt 2>&1
tail -5 $O/pytest.txt
for p in 0 3 4; do
for shape in "1 128 128" "2 128 128" "3 128 128" "1 64 64" "2 256 128"; do
  echo "FPCC_CONV_PERSIST=$p" >> $O/persist.txt
  FPCC_CONV_PERSIST=$p ONLY=pattern timeout 300 python3 tools/conv_probe.py $shape 20 >> $O/persist.txt 2>&1
done
done
grep -v amdgpu $O/persist.txt
