#!/bin/bash
# rocprofv3 kernel stats of the secondary configurations (cfg#3 integer codec, cfg#4 colour codec, cfg#5 training step)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r02s2; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --stats -d $O/int -o s --output-format csv -- python3 tools/timeline_int.py > $O/int.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/color -o s --output-format csv -- python3 tools/timeline_color.py > $O/color.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats -d $O/train -o s --output-format csv -- python3 bench_train.py --steps 4 --warmup 1 > $O/train.log 2>&1
tail -3 $O/int.log; tail -3 $O/color.log; tail -2 $O/train.log
find $O -name '*kernel_stats.csv' | xargs ls -la
