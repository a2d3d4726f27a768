#!/bin/bash
# rocprofv3 evidence for profiles/r03: kernel stats of the bench command, then FETCH_SIZE / WRITE_SIZE in separate PMC passes
# usage: bash tools/r03/profile.sh <tag> [stats|all]
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
TAG=${1:-a}; WHAT=${2:-all}
O=gpurun_out/r03p_$TAG; mkdir -p $O
CMD="python3 bench.py --steps 3 --warmup 1 --cpu-baseline 0 --secondary 0"
timeout 300 rocprofv3 --kernel-trace --stats -d $O/stats -o s --output-format csv -- $CMD > $O/stats.log 2>&1
if [ "$WHAT" = "all" ]; then
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch -o p --output-format csv -- $CMD > $O/fetch.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write -o p --output-format csv -- $CMD > $O/write.log 2>&1
fi
find $O -name '*.csv' | xargs ls -la | head -30
python profiles/summarize.py $(find $O/stats -name "*kernel_stats.csv" | head -1) 5 | head -60
