#!/bin/bash
# mid-round look at the default bench (batch 4, 2 batches in flight): kernel stats, per-launch conv table, stream gaps
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r05M; mkdir -p $O
timeout 600 python3 bench.py --steps 10 --warmup 3 --secondary 0 --cpu-baseline 0 --dump-trace $O/conv_launches.txt > $O/bench.json 2> $O/bench.err; head -c 600 $O/bench.json; echo
CMD="python3 bench.py --steps 3 --warmup 1 --cpu-baseline 0 --secondary 0"
timeout 400 rocprofv3 --kernel-trace --stats -d $O/stats -o s --output-format csv -- $CMD > $O/stats.log 2>&1
timeout 400 rocprofv3 --kernel-trace -d $O/gap -o p --output-format csv -- python3 bench.py --steps 8 --warmup 2 --cpu-baseline 0 --secondary 0 > $O/gap.log 2>&1
python3 profiles/step_gaps.py $(find $O/gap -name p_kernel_trace.csv | head -1) "batch 4, two batches in flight" > $O/step_gaps.md 2>&1
head -30 $O/step_gaps.md
python3 profiles/summarize.py $(find $O/stats -name s_kernel_stats.csv | head -1) 1 > $O/summary.md 2>&1; head -50 $O/summary.md
python3 profiles/conv_by_level.py $O/conv_launches.txt > $O/conv_by_level.md 2>&1; cat $O/conv_by_level.md | head -60
find $O -name '*_kernel_trace.csv' -delete
