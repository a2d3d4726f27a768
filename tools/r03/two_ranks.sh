#!/bin/bash
# the N > 1 path of bench.py (replica timing + the DDP step of cfg#5) as two ranks on the box's one GPU over gloo: a dry run of the
# code path the driver's multi-GPU command takes, not a measurement
O=gpurun_out/r03_two; mkdir -p $O
export FPCC_BENCH_ONE_DEVICE=1 FPCC_BENCH_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 4 --warmup 1 --ddp-steps 3 > $O/out.txt 2> $O/err.txt
echo "rc=$?"; tail -c 3000 $O/out.txt; grep -v amdgpu.ids $O/err.txt | tail -15
