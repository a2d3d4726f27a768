#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r03_sq; mkdir -p $O
for lvl in 1 2; do
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 -d $O/l$lvl -o p --output-format csv -- python3 tools/conv_probe.py $lvl 128 128 5 > $O/l$lvl.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE -d $O/m$lvl -o p --output-format csv -- python3 tools/conv_probe.py $lvl 128 128 5 > $O/m$lvl.log 2>&1
done
python - <<'PY'
import csv, collections, glob
for d in sorted(glob.glob('gpurun_out/r03_sq/*/')):
    f = glob.glob(d + '*counter_collection.csv')
    if not f: continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f[0])):
        if 'k_conv_wave' not in r['Kernel_Name']: continue
        acc[r['Kernel_Name'][:70]][r['Counter_Name']] += float(r['Counter_Value'])
        cnt[(r['Kernel_Name'][:70], r['Counter_Name'])] += 1
    for k, v in acc.items():
        print(d, k)
        for c, val in sorted(v.items()):
            print('   ', c, val / cnt[(k, c)])
PY
