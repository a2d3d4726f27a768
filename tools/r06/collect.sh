#!/bin/bash
# gpurun_out/r06F (made by tools/r05/final.sh on the GPU box) -> profiles/r06/final_*   (run in the build container)
S=gpurun_out/r06F; D=profiles/r06
f() { find $S/$1 -name "$2" | head -1; }
cp $(f stats s_kernel_stats.csv) $D/final_kernel_stats.csv
FRAMES=${1:-116}
{ echo "# rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 3 --warmup 1 --latency-frames 2 --cpu-baseline 0 --secondary 0   (round 6, final state; batch 16, two batches in flight: 16 reference frames + 4 one-frame-at-a-time frames + 2 warm-up and 3 timed batches of 16 + 1 clock-probe batch of 16 = $FRAMES frame-equivalents)"; python profiles/summarize.py $D/final_kernel_stats.csv $FRAMES; } > $D/final_summary.md
cp $S/pmc_traffic.json $D/final_pmc_traffic.json; cp $S/pmc_traffic.md $D/final_pmc_traffic.md       # summarised on the GPU box (final.sh)
python - <<PY
import json
p = '$D/final_pmc_traffic.json'
d = json.load(open(p))
d['_meta'] = {'frame_equivalents': $FRAMES, 'command': 'bench.py --steps 3 --warmup 1 --latency-frames 2 --cpu-baseline 0 --secondary 0 (batch 16, two batches in flight)'}
json.dump(d, open(p, 'w'), indent=1)
PY
cp $S/hbm_bandwidth.md $D/final_hbm_bandwidth.md
{ echo; echo "## The helper kernels by launch size (profiles/hbm_bandwidth_by_size.py)"; echo; cat $S/hbm_bandwidth_by_size.md; } >> $D/final_hbm_bandwidth.md
cp $S/conv_launches.txt $D/final_conv_launches.txt
python profiles/conv_by_level.py $D/final_conv_launches.txt > $D/final_conv_by_level.md
cp $S/bench_default.json $D/final_bench.json
for c in b1_d1 b1_d2 b8_d2 b16_d1; do cp $S/bench_$c.json $D/final_bench_$c.json; done
cp $S/mfma_busy.md $D/final_mfma_busy.md
cp $S/step_gaps.md $D/final_step_gaps.md
for c in int color train; do cp $(f $c s_kernel_stats.csv) $D/final_${c}_kernel_stats.csv; done
{ echo "# rocprofv3 --kernel-trace --stats -- integer codec (cfg#3): tools/timeline_int.py = 5 x (compress + decompress) of the 113 108-voxel LiDAR-like frame   (round 6, final state)"; python profiles/summarize.py $D/final_int_kernel_stats.csv 5; } > $D/final_int_summary.md
{ echo "# rocprofv3 --kernel-trace --stats -- colour codec (cfg#4): tools/timeline_color.py   (round 6, final state)"; python profiles/summarize.py $D/final_color_kernel_stats.csv 5; } > $D/final_color_summary.md
{ echo "# rocprofv3 --kernel-trace --stats -- training step (cfg#5): bench_train.py --steps 4 --warmup 1, 8 clouds per step   (round 6, final state; 6 steps incl. warm-up and the consensus probe)"; python profiles/summarize.py $D/final_train_kernel_stats.csv 6; } > $D/final_train_summary.md
tail -1 $S/pytest_gpu.txt; tail -3 $S/pytest_gpu.txt | head -1; tail -1 $S/smoke.txt
