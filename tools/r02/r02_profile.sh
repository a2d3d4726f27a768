#!/bin/bash
# rocprofv3 evidence for profiles/r02: kernel stats of the bench command, then FETCH_SIZE / WRITE_SIZE in separate PMC passes
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r02p; mkdir -p $O
CMD="python3 bench.py --steps 3 --warmup 1 --cpu-baseline 0 --secondary 0"
timeout 300 rocprofv3 --kernel-trace --stats -d $O/stats -o s --output-format csv -- $CMD > $O/stats.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch -o p --output-format csv -- $CMD > $O/fetch.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write -o p --output-format csv -- $CMD > $O/write.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 -d $O/sq -o p --output-format csv -- $CMD > $O/sq.log 2>&1
find $O -name '*.csv' | xargs ls -la | head -30
python profiles/summarize.py $(find $O/stats -name '*kernel_stats.csv' | head -1) 4 | head -45
