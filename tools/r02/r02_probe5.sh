#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r02g; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_rans_dev.py tests/test_gpu_me_api.py -x -q > $O/pytest_new.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_new.txt
tail -15 $O/pytest_new.txt
run() { timeout 300 python tools/conv_probe.py "$@" 2>&1 | grep -v amdgpu.ids; }
# tiled kernel: where does its time go (dbg 1 = no gather traffic, 2 = W from one chunk, 4 = no barrier, 7 = all)
for lvl in 1 2 3; do
  for tile in 1 2; do
    FPCC_MFMA_TILE=$tile ONLY=pattern NBWS=0 DBG=0,1,2,4,7 run $lvl 128 128 20 | sed "s/tiled kernel/tiled kernel tile=$tile/" | tee -a $O/tiled_dbg.txt
  done
done
timeout 300 python tools/device_rans_ab.py > $O/device_rans.md 2> $O/device_rans.err; tail -3 $O/device_rans.err; cat $O/device_rans.md
python tools/make_gpu_golden.py $O/v2_stream.json
( time timeout 420 python bench.py ) > $O/bench_default.json 2> $O/bench_default.err
tail -4 $O/bench_default.err
python -c "
import json
d = json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'])
print(json.dumps(d['config']['secondary'], indent=1))
print(json.dumps(d['cpu_baseline'], indent=1))"
timeout 300 python -m pytest tests/test_gpu_codec_int.py tests/test_gpu_int_ops.py tests/test_gpu_ptq.py -x -q 2>&1 | tail -3
