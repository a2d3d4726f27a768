#!/bin/bash
# SQ counters of the convolution family in the bench command: are the waves of the dominant kernel parked (waiting for operands) or issue-stalled?
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r06Q; rm -rf $O; mkdir -p $O
CMD="python3 bench.py --steps 3 --warmup 1 --latency-frames 2 --cpu-baseline 0 --secondary 0"
timeout 500 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_WAIT_INST_LDS -d $O/sq -o p --output-format csv -- $CMD > $O/sq.log 2>&1
python3 tools/r06/sq_summary.py $(find $O/sq -name "*counter_collection.csv" | head -1) > $O/conv_sq.md 2>&1
timeout 500 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA -d $O/sq2 -o p --output-format csv -- $CMD > $O/sq2.log 2>&1
python3 tools/r06/sq_summary.py $(find $O/sq2 -name "*counter_collection.csv" | head -1) > $O/conv_sq2.md 2>&1
rm -rf $O/sq $O/sq2
head -12 $O/conv_sq.md; head -8 $O/conv_sq2.md
