#!/bin/bash
# GPU session 3 (round 4): k_conv_lds without the LDS neighbour table (3 workgroups per CU at 2 row blocks): parity, timing, SQ counters
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r04c; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_conv.py -x -q -k "lds_operand" > $O/pytest.txt 2>&1
tail -3 $O/pytest.txt
for shape in "1 128 128" "2 128 128" "1 64 64" "2 256 128" "1 128 64" "1 64 128"; do
  ONLY=pattern LDS=0,2,3,4 timeout 300 python3 tools/conv_probe.py $shape 20 >> $O/lds.txt 2>&1
done
grep -v amdgpu $O/lds.txt
pmc() {  # name, counters...
  n=$1; shift
  ONLY=pattern LDS=0,2,4 timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d $O/pmc_$n -o p --output-format csv -- python3 tools/conv_probe.py 1 128 128 3 > $O/pmc_$n.log 2>&1
}
pmc sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32
pmc sq2 SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VALU GRBM_GUI_ACTIVE
pmc sq3 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SALU SQ_INST_CYCLES_VMEM_RD SQ_VALU_MFMA_COEXEC_CYCLES SQ_LEVEL_WAVES
python3 - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob('gpurun_out/r04c/pmc_*/')):
    for f in glob.glob(d + '**/*counter_collection.csv', recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            if 'k_conv_' not in r['Kernel_Name']: continue
            acc[r['Kernel_Name'][:90]][r['Counter_Name']].append(float(r['Counter_Value']))
        for k, v in acc.items():
            print(d, k)
            for c, vals in sorted(v.items()):
                print('   ', c, 'mean %.4g over %d' % (sum(vals) / len(vals), len(vals)))
PY
