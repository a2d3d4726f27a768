#!/bin/bash
# gpurun_out/r03z (made by tools/r03/final.sh on the GPU box) -> profiles/r03/final_*   (run in the build container)
S=gpurun_out/r03z; D=profiles/r03
cp $S/stats/s_kernel_stats.csv $D/final_kernel_stats.csv
{ echo "# rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 3 --warmup 1 --cpu-baseline 0 --secondary 0   (round 3, final state; 5 steps: warm-up, 3 timed, 1 clock-probe step)"; python profiles/summarize.py $S/stats/s_kernel_stats.csv 5; } > $D/final_summary.md
python profiles/pmc_summary.py $S/fetch/p_counter_collection.csv $S/write/p_counter_collection.csv $D/final_pmc_traffic.json > $D/final_pmc_traffic.md
python profiles/hbm_bandwidth.py $S/fetch $S/write > $D/final_hbm_bandwidth.md
cp $S/conv_launches.txt $D/final_conv_launches.txt
python profiles/conv_by_level.py $D/final_conv_launches.txt > $D/final_conv_by_level.md
cp $S/bench_default.json $D/final_bench.json
for c in int color train; do cp $S/$c/s_kernel_stats.csv $D/final_${c}_kernel_stats.csv; done
{ echo "# rocprofv3 --kernel-trace --stats -- integer codec (cfg#3): tools/timeline_int.py = 5 x (compress + decompress) of the 113 108-voxel LiDAR-like frame   (round 3, final state)"; python profiles/summarize.py $S/int/s_kernel_stats.csv 5; } > $D/final_int_summary.md
{ echo "# rocprofv3 --kernel-trace --stats -- colour codec (cfg#4): tools/timeline_color.py   (round 3, final state)"; python profiles/summarize.py $S/color/s_kernel_stats.csv 5; } > $D/final_color_summary.md
{ echo "# rocprofv3 --kernel-trace --stats -- training step (cfg#5): bench_train.py --steps 4 --warmup 1, 8 clouds per step   (round 3, final state)"; python profiles/summarize.py $S/train/s_kernel_stats.csv 5; } > $D/final_train_summary.md
tail -1 $S/pytest_gpu.txt; tail -3 $S/pytest_gpu.txt | head -1
{ echo; echo "## The helper kernels by launch size (profiles/hbm_bandwidth_by_size.py)"; echo; python profiles/hbm_bandwidth_by_size.py $S/fetch $S/write; } >> $D/final_hbm_bandwidth.md
