#!/bin/bash
# GPU session 1 (round 4): counter list, stage stamps of the small levels, A-whole-stage variant, L1/L2 counters of the folded kernel
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r04a; mkdir -p $O
rocprofv3 -L > $O/counters.txt 2>&1
timeout 600 python3 tools/r04/stamp_probe.py 4,5,6 > $O/stamps.txt 2>&1
for lv in 1 2; do
  FPCC_GROUPED_FOLD_ROWS=1 ONLY=pattern DBG=0,32,0,32 timeout 300 python3 tools/conv_probe.py $lv 128 128 20 >> $O/astage.txt 2>&1
done
FPCC_GROUPED_FOLD_ROWS=1 ONLY=pattern DBG=0,32,0,32 timeout 300 python3 tools/conv_probe.py 1 64 64 20 >> $O/astage.txt 2>&1
FPCC_GROUPED_FOLD_ROWS=1 ONLY=pattern DBG=0,32 timeout 300 python3 tools/conv_probe.py 1 256 128 20 >> $O/astage.txt 2>&1
pmc() {  # name, counters...
  n=$1; shift
  FPCC_GROUPED_FOLD_ROWS=1 ONLY=pattern DBG=0,32 timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d $O/pmc_$n -o p --output-format csv -- python3 tools/conv_probe.py 1 128 128 3 > $O/pmc_$n.log 2>&1
}
pmc l1a TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
pmc l1b TCP_TOTAL_ACCESSES_sum TCP_TOTAL_READ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TA_TCP_STATE_READ_sum
pmc ta TA_BUSY_avr TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUFFER_WAVEFRONTS_sum TA_FLAT_READ_WAVEFRONTS_sum
pmc l2 TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_READ_sum
pmc ea TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum GRBM_GUI_ACTIVE
python3 - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob('gpurun_out/r04a/pmc_*/')):
    for f in glob.glob(d + '**/*counter_collection.csv', recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            if 'k_conv_wave' not in r['Kernel_Name']: continue
            acc[r['Kernel_Name'][40:110]][r['Counter_Name']].append(float(r['Counter_Value']))
        for k, v in acc.items():
            print(d, k)
            for c, vals in sorted(v.items()):
                print('   ', c, 'mean %.4g over %d' % (sum(vals) / len(vals), len(vals)))
PY
