#!/bin/bash
# GPU session 24: int8 tiled convolution alone, per level, with the timing ablations
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r04z; mkdir -p $O
for d in 0 1 7; do
  echo "== FPCC_I8_DBG=$d"; FPCC_I8_DBG=$d timeout 300 python3 tools/r04/i8_probe.py > $O/i8_dbg$d.txt 2>&1; grep -v amdgpu.ids $O/i8_dbg$d.txt
done
