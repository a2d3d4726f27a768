#!/bin/bash
# evidence of the round's final state: full GPU suite + smoke + default bench (with the per-launch table), the one-frame-at-a-time and
# one-batch-at-a-time benches, rocprofv3 kernel stats and FETCH_SIZE / WRITE_SIZE / MFMA-busy passes of the bench command, kernel traces
# for the gap timelines, kernel stats of the secondary configurations
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r06F; mkdir -p $O
timeout 1700 python3 -m pytest tests -m gpu -x -q --durations=5 > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt
tail -4 $O/pytest_gpu.txt
timeout 120 python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?" >> $O/smoke.txt; tail -2 $O/smoke.txt
timeout 900 python3 bench.py --dump-trace $O/conv_launches.txt > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; head -c 500 $O/bench_default.json; echo
for cfg in "1 1" "1 2" "8 2" "16 1"; do set -- $cfg
  timeout 400 python3 bench.py --batch $1 --frames-in-flight $2 --cpu-baseline 0 --secondary 0 > $O/bench_b$1_d$2.json 2> $O/bench_b$1_d$2.err; head -c 200 $O/bench_b$1_d$2.json; echo
done
CMD="python3 bench.py --steps 3 --warmup 1 --latency-frames 2 --cpu-baseline 0 --secondary 0"
timeout 400 rocprofv3 --kernel-trace --stats -d $O/stats -o s --output-format csv -- $CMD > $O/stats.log 2>&1
timeout 500 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch -o p --output-format csv -- $CMD > $O/fetch.log 2>&1
timeout 500 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write -o p --output-format csv -- $CMD > $O/write.log 2>&1
timeout 500 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 -d $O/busy -o p --output-format csv -- $CMD > $O/busy.log 2>&1
python3 profiles/mfma_busy.py $O/busy > $O/mfma_busy.md 2>&1; head -8 $O/mfma_busy.md
timeout 400 rocprofv3 --kernel-trace -d $O/gap -o p --output-format csv -- python3 bench.py --steps 10 --warmup 2 --latency-frames 1 --cpu-baseline 0 --secondary 0 > $O/gap.log 2>&1
python3 profiles/step_gaps.py $(find $O/gap -name p_kernel_trace.csv | head -1) "batches of 16 frames, two batches in flight (bench.py --steps 10 --warmup 2 --latency-frames 1, the default batch and depth; the window starts behind the 19 single frames the bench codes first)" 0.66 0.97 > $O/step_gaps.md 2>&1
head -12 $O/step_gaps.md
timeout 300 rocprofv3 --kernel-trace --stats -d $O/int -o s --output-format csv -- python3 tools/timeline_int.py > $O/int.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/color -o s --output-format csv -- python3 tools/timeline_color.py > $O/color.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats -d $O/train -o s --output-format csv -- python3 bench_train.py --steps 4 --warmup 1 > $O/train.log 2>&1
tail -3 $O/int.log; tail -3 $O/color.log
find $O -name 's_kernel_trace.csv' -delete; rm -f $O/write/p_kernel_trace.csv; find $O/gap $O/busy -name p_kernel_trace.csv -delete
# what travels back must stay under 64 MiB: the counter tables are summarised here and only their gzip goes home
python3 profiles/pmc_summary.py $(find $O/fetch -name p_counter_collection.csv | head -1) $(find $O/write -name p_counter_collection.csv | head -1) $O/pmc_traffic.json > $O/pmc_traffic.md 2>&1
FD=$(dirname $(find $O/fetch -name p_counter_collection.csv | head -1)); WD=$(dirname $(find $O/write -name p_counter_collection.csv | head -1))
python3 profiles/hbm_bandwidth.py $FD $WD > $O/hbm_bandwidth.md 2>&1
python3 profiles/hbm_bandwidth_by_size.py $FD $WD > $O/hbm_bandwidth_by_size.md 2>&1
find $O/fetch -name p_kernel_trace.csv -delete
find $O/fetch $O/write $O/busy -name p_counter_collection.csv -exec gzip -9 {} \;
du -sh gpurun_out; find $O -name '*.csv*' | xargs ls -la | head -40
# round 6: the helper kernels alone (time, algorithmic bytes) and their SQ counters
python3 tools/r06/helper_bench.py 16 > $O/helpers.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU -d $O/sq -o p --output-format csv -- python3 tools/r06/helper_bench.py 8 > $O/sq.log 2>&1
python3 tools/r06/sq_summary.py $(find $O/sq -name "*counter_collection.csv" | head -1) > $O/helpers_sq.md 2>&1; rm -rf $O/sq
du -sh gpurun_out
