#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
CPU_OPS=1 TOP=70 timeout 300 python3 tools/train_launches.py > $O/g5_train_launches.log 2>&1; tail -120 $O/g5_train_launches.log
timeout 200 python3 tools/r05/union_fine.py > $O/g5_union_fine.log 2>&1; cat $O/g5_union_fine.log
timeout 400 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVES -d $O/i8pmc -o p --output-format csv -- python3 tools/r05/i8_probe.py 4 > $O/g5_i8pmc.log 2>&1
tail -12 $O/g5_i8pmc.log
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/r05/i8pmc/**/p_counter_collection.csv', recursive=True)
print(f)
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for row in csv.DictReader(open(f[0])):
    k = row['Kernel_Name'][:60]
    agg[k][row['Counter_Name']] += float(row['Counter_Value'])
    cnt[k] += 1
for k, v in agg.items():
    if 'conv_i8' in k:
        print(k, dict(v), 'rows', cnt[k])
PY
find $O/i8pmc -name '*kernel_trace.csv' -delete
