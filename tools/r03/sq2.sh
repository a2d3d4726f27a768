#!/bin/bash
# SQ counters of the final kernels on one 27-offset 128 -> 128 layer, pattern order only: folded (level 1 default), four-wave (level 1 with
# FPCC_GROUPED_FOLD_ROWS=0, level 2 default)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r03_sq2; mkdir -p $O
export ONLY=pattern WARM_MS=5
run() {  # tag level env
  env_fold=$3
  FPCC_GROUPED_FOLD_ROWS=$env_fold timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 -d $O/a_$1 -o p --output-format csv -- python3 tools/conv_probe.py $2 128 128 5 > $O/a_$1.log 2>&1
  FPCC_GROUPED_FOLD_ROWS=$env_fold timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE -d $O/b_$1 -o p --output-format csv -- python3 tools/conv_probe.py $2 128 128 5 > $O/b_$1.log 2>&1
}
run l1_folded 1 102400
run l1_fourwave 1 0
run l2_fourwave 2 102400
python3 - <<'PY'
import csv, collections, glob
tab = collections.OrderedDict()
for d in sorted(glob.glob('gpurun_out/r03_sq2/*/')):
    f = glob.glob(d + '*counter_collection.csv')
    if not f: continue
    tag = d.rstrip('/').split('/')[-1][2:]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if 'k_conv_wave' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
    for c, v in acc.items():
        tab.setdefault(c, {})[tag] = sum(v) / len(v)
tags = ['l1_folded', 'l1_fourwave', 'l2_fourwave']
with open('gpurun_out/r03_sq2/table.md', 'w') as out:
    out.write('| counter (mean per launch, summed over XCDs) | ' + ' | '.join(tags) + ' |\n|---|' + '---:|' * len(tags) + '\n')
    for c in sorted(tab):
        out.write(f'| {c} | ' + ' | '.join(f'{tab[c].get(t, float("nan")):.4g}' for t in tags) + ' |\n')
print(open('gpurun_out/r03_sq2/table.md').read())
PY
