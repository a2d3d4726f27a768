#!/bin/bash
# GPU session 11: kernel stats of the step at depth 1 and depth 2 (mid-round snapshot)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r04m; mkdir -p $O
CMD="python3 bench.py --steps 6 --warmup 2 --cpu-baseline 0 --secondary 0"
timeout 400 rocprofv3 --kernel-trace --stats -d $O/stats1 -o s --output-format csv -- $CMD --frames-in-flight 1 > $O/stats1.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats -d $O/stats2 -o s --output-format csv -- $CMD --frames-in-flight 2 > $O/stats2.log 2>&1
timeout 300 python3 bench.py --steps 20 --warmup 5 --cpu-baseline 0 --secondary 0 --frames-in-flight 1 --dump-trace $O/conv_launches.txt > $O/bench1.json 2> $O/bench1.err
rm -f $O/stats2/s_kernel_trace.csv
ls -la $O/stats1 | head
