#!/bin/bash
# matrix-pipe busy fraction per kernel inside the bench step (one PMC pass)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r03_busy; mkdir -p $O
timeout 400 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 -d $O/p -o p --output-format csv -- python3 bench.py --steps 3 --warmup 1 --cpu-baseline 0 --secondary 0 > $O/log.txt 2>&1
python3 profiles/mfma_busy.py $O/p | tee $O/busy.md
