#!/bin/bash
O=gpurun_out/r03_w22; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_conv.py -x -q -m gpu > $O/pytest.txt 2>&1; echo "pytest rc $?" >> $O/pytest.txt
tail -4 $O/pytest.txt
for lvl in 1 2 3; do
  for shape in "128 128" "64 64" "256 128"; do
    ONLY=pattern W22=1 timeout 200 python tools/conv_probe.py $lvl $shape 20 2>&1 | grep -v amdgpu.ids | tee -a $O/probe.txt
  done
done
