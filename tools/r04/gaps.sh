#!/bin/bash
# the two gap timelines of tools/r04/final.sh alone (kernel traces of the bench, one frame at a time / two frames in flight)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r04F; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace -d $O/gap1 -o p --output-format csv -- python3 bench.py --steps 12 --warmup 2 --cpu-baseline 0 --secondary 0 --frames-in-flight 1 > $O/gap1.log 2>&1
timeout 300 rocprofv3 --kernel-trace -d $O/gap2 -o p --output-format csv -- python3 bench.py --steps 20 --warmup 2 --cpu-baseline 0 --secondary 0 --frames-in-flight 2 > $O/gap2.log 2>&1
python3 profiles/step_gaps.py $(find $O/gap1 -name p_kernel_trace.csv | head -1) "one frame at a time (bench.py --steps 12 --warmup 2 --frames-in-flight 1)" > $O/step_gaps_1.md
python3 profiles/step_gaps.py $(find $O/gap2 -name p_kernel_trace.csv | head -1) "two frames in flight (bench.py --steps 20 --warmup 2, the default depth)" > $O/step_gaps_2.md
cat $O/step_gaps_1.md $O/step_gaps_2.md
find $O/gap1 $O/gap2 -name p_kernel_trace.csv -delete
