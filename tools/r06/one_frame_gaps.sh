#!/bin/bash
# one frame at a time (the BASELINE definition of the metric): where the GPU idles
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r06/g1; rm -rf $O; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace -d $O/gap -o p --output-format csv -- python3 bench.py --batch 1 --frames-in-flight 1 --steps 16 --warmup 2 --latency-frames 1 --cpu-baseline 0 --secondary 0 > $O/gap.log 2>&1
tail -2 $O/gap.log | cut -c1-400
python3 profiles/step_gaps.py $(find $O/gap -name p_kernel_trace.csv | head -1) "one frame at a time (bench.py --batch 1 --frames-in-flight 1 --steps 16 --warmup 2 --latency-frames 1)" > $O/step_gaps_one_frame.md 2>&1
cat $O/step_gaps_one_frame.md
python3 - $(find $O/gap -name p_kernel_trace.csv | head -1) $(find $O/gap -name p_memory_copy_trace.csv | head -1) $O <<'PY'
import csv, sys, os
k, c, O = sys.argv[1:4]
with open(os.path.join(O, 'kernel_compact.csv'), 'w') as out:
    for r in csv.DictReader(open(k)):
        out.write(f"{r['Start_Timestamp']},{r['End_Timestamp']},{r['Kernel_Name'][:70].replace(',', ';')}\n")
with open(os.path.join(O, 'copy_compact.csv'), 'w') as out:
    for r in csv.DictReader(open(c)):
        out.write(f"{r['Start_Timestamp']},{r['End_Timestamp']},{r.get('Direction', '')},{r.get('Bytes', r.get('Size', ''))}\n")
PY
rm -rf $O/gap; ls -la $O
