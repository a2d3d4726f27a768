#!/bin/bash
# evidence of the round's final state: full GPU suite + smoke + default bench (with the per-launch table), rocprofv3 kernel stats and
# FETCH_SIZE / WRITE_SIZE passes of the bench command, kernel stats of the secondary configurations
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r03z; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q --durations=5 > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt
tail -10 $O/pytest_gpu.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?" >> $O/smoke.txt; tail -2 $O/smoke.txt
timeout 480 python bench.py --dump-trace $O/conv_launches.txt > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; head -c 600 $O/bench_default.json; echo
CMD="python3 bench.py --steps 3 --warmup 1 --cpu-baseline 0 --secondary 0"
timeout 300 rocprofv3 --kernel-trace --stats -d $O/stats -o s --output-format csv -- $CMD > $O/stats.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch -o p --output-format csv -- $CMD > $O/fetch.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write -o p --output-format csv -- $CMD > $O/write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/int -o s --output-format csv -- python3 tools/timeline_int.py > $O/int.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/color -o s --output-format csv -- python3 tools/timeline_color.py > $O/color.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats -d $O/train -o s --output-format csv -- python3 bench_train.py --steps 4 --warmup 1 > $O/train.log 2>&1
tail -3 $O/int.log; tail -3 $O/color.log
rm -f $O/int/s_kernel_trace.csv $O/color/s_kernel_trace.csv $O/train/s_kernel_trace.csv $O/write/p_kernel_trace.csv 2>/dev/null
find $O -name '*.csv' | xargs ls -la | head -40
