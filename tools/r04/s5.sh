#!/bin/bash
# GPU session 5 (round 4): epilogue loads hoisted above the stores + row-major neighbour table in the prologue: parity, then timing
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r04f; mkdir -p $O
timeout 1500 python3 -m pytest tests/test_gpu_conv.py tests/test_gpu_mlp_chain.py -x -q > $O/pytest.txt 2>&1
tail -5 $O/pytest.txt
for shape in "1 128 128" "2 128 128" "3 128 128" "4 128 128" "1 64 64" "2 256 128"; do
  ONLY=pattern LDS=0,2,4 timeout 300 python3 tools/conv_probe.py $shape 20 >> $O/lds.txt 2>&1
done
grep -v amdgpu $O/lds.txt
python3 tools/r04/lds_stamp_probe.py 1 128 128 2 > $O/lds_stamps.txt 2>&1; grep -v amdgpu $O/lds_stamps.txt
