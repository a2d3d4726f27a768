#!/bin/bash
# GPU session 30: training step now (cfg#5) + kernel stats; helper-kernel bandwidth table of the bench command (FETCH/WRITE passes)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r04A; mkdir -p $O
timeout 400 python3 bench_train.py --steps 8 --warmup 3 > $O/train.json 2> $O/train.err; cat $O/train.json | head -c 600; echo
timeout 400 rocprofv3 --kernel-trace --stats -d $O/train -o s --output-format csv -- python3 bench_train.py --steps 4 --warmup 1 > $O/train.log 2>&1
python3 profiles/summarize.py $O/train/s_kernel_stats.csv 5 | head -45
rm -f $O/train/s_kernel_trace.csv
