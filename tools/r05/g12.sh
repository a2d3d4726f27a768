#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
for d in 0 8 32 64; do echo "== i8 probe FPCC_I8_DBG=$d"; FPCC_I8_DBG=$d timeout 200 python3 tools/r05/i8_probe.py 20 2>&1 | grep "^|" | cut -d'|' -f2,3,6 | tee $O/g12_i8_dbg$d.txt; done
