#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r02pw; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_conv.py -x -q -k "pointwise" > $O/pytest.txt 2>&1; echo "pytest rc=$?" >> $O/pytest.txt
tail -5 $O/pytest.txt
for a in "272431 128 128" "272431 64 128" "272431 256 128" "272431 64 64" "70509 128 128" "70509 256 128" "997645 32 32" "35000 128 128"; do timeout 100 python tools/pointwise_probe.py $a 2>&1 | grep -v amdgpu | grep -v "no \|neither"; done
timeout 100 python tools/pointwise_probe.py 272431 128 128 2>&1 | grep "no \|neither"
