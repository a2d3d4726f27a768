#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r03_train2; mkdir -p $O
timeout 400 rocprofv3 --kernel-trace --stats -d $O/train -o s --output-format csv -- python3 bench_train.py --steps 4 --warmup 1 > $O/train.log 2>&1
tail -1 $O/train.log | cut -c1-200
python3 profiles/summarize.py $O/train/s_kernel_stats.csv 5 | head -24
rm -f $O/train/s_kernel_trace.csv
