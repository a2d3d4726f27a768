#!/bin/bash
# gpurun_out/r04F (made by tools/r04/final.sh on the GPU box) -> profiles/r04/final_*   (run in the build container)
S=gpurun_out/r04F; D=profiles/r04
f() { find $S/$1 -name "$2" | head -1; }
cp $(f stats s_kernel_stats.csv) $D/final_kernel_stats.csv
{ echo "# rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 3 --warmup 1 --cpu-baseline 0 --secondary 0   (round 4, final state; 10 frames: 1 serial + 2 pipelined warm-up, 3 timed, 3 latency frames, 1 clock-probe frame)"; python profiles/summarize.py $D/final_kernel_stats.csv 10; } > $D/final_summary.md
python profiles/pmc_summary.py $(f fetch p_counter_collection.csv) $(f write p_counter_collection.csv) $D/final_pmc_traffic.json > $D/final_pmc_traffic.md
python profiles/hbm_bandwidth.py $(dirname $(f fetch p_counter_collection.csv)) $(dirname $(f write p_counter_collection.csv)) > $D/final_hbm_bandwidth.md
{ echo; echo "## The helper kernels by launch size (profiles/hbm_bandwidth_by_size.py)"; echo; python profiles/hbm_bandwidth_by_size.py $(dirname $(f fetch p_counter_collection.csv)) $(dirname $(f write p_counter_collection.csv)); } >> $D/final_hbm_bandwidth.md
cp $S/conv_launches.txt $D/final_conv_launches.txt
python profiles/conv_by_level.py $D/final_conv_launches.txt > $D/final_conv_by_level.md
cp $S/bench_default.json $D/final_bench.json
cp $S/bench_depth1.json $D/final_bench_one_frame_at_a_time.json
cp $S/mfma_busy.md $D/final_mfma_busy.md
for c in int color train; do cp $(f $c s_kernel_stats.csv) $D/final_${c}_kernel_stats.csv; done
{ echo "# rocprofv3 --kernel-trace --stats -- integer codec (cfg#3): tools/timeline_int.py = 5 x (compress + decompress) of the 113 108-voxel LiDAR-like frame   (round 4, final state)"; python profiles/summarize.py $D/final_int_kernel_stats.csv 5; } > $D/final_int_summary.md
{ echo "# rocprofv3 --kernel-trace --stats -- colour codec (cfg#4): tools/timeline_color.py   (round 4, final state)"; python profiles/summarize.py $D/final_color_kernel_stats.csv 5; } > $D/final_color_summary.md
{ echo "# rocprofv3 --kernel-trace --stats -- training step (cfg#5): bench_train.py --steps 4 --warmup 1, 8 clouds per step   (round 4, final state; 6 steps incl. warm-up and the consensus probe)"; python profiles/summarize.py $D/final_train_kernel_stats.csv 6; } > $D/final_train_summary.md
tail -1 $S/pytest_gpu.txt; tail -3 $S/pytest_gpu.txt | head -1; tail -1 $S/smoke.txt
