#!/usr/bin/env python3
"""DDP training throughput of lossy_coord_v2/baseline_r1 on synthetic ShapeNet-like batches (BASELINE.json configs[4]).

    python bench_train.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
        bench_train.py --gpus N --steps K --warmup W

The global batch of 8 clouds is split over the ranks (strong scaling); one rank per GPU, gradients all-reduced over RCCL.
Prints ONE JSON line on rank 0.  (bench.py remains the headline inference benchmark.)
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--resolution', type=int, default=128)
    args = ap.parse_args()
    from fastpcc_amd.train import bench
    out = bench(args.steps, args.warmup, args.gpus, args.resolution)
    if out is not None:
        print(json.dumps(out))
    import torch.distributed as dist
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
