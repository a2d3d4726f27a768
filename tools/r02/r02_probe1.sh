#!/bin/bash
# round-2 first look: counter list, baseline bench, SQ counters of one 70K-row and one 272K-row 128->128 3x3x3 layer
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r02a; mkdir -p $O
rocprofv3 -L > $O/counters.txt 2>&1
python bench.py --steps 10 --warmup 3 --cpu-baseline 0 > $O/bench.json 2> $O/bench.err
for lvl in 1 2 3; do
  python tools/conv_probe.py $lvl 128 128 20 > $O/probe_l$lvl.txt 2>&1
done
for lvl in 1 2; do
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 -d $O/pmc_sq_l$lvl -o p --output-format csv -- python tools/conv_probe.py $lvl 128 128 3 > $O/pmc_sq_l$lvl.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $O/pmc_sq2_l$lvl -o p --output-format csv -- python tools/conv_probe.py $lvl 128 128 3 > $O/pmc_sq2_l$lvl.log 2>&1
  rocprofv3 --kernel-trace --pmc TA_TA_BUSY_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TA_TCP_STATE_READ_sum -d $O/pmc_ta_l$lvl -o p --output-format csv -- python tools/conv_probe.py $lvl 128 128 3 > $O/pmc_ta_l$lvl.log 2>&1
done
find $O -name '*.csv' -size +20M -delete
ls -R $O | head -50
