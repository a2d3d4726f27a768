#!/usr/bin/env python3
"""Integer codec (cfg#3): sweeps (or batches of sweeps) in flight on one GPU (serving.FramePipeline: D codec contexts over one set of
weights on one stream) -- does another frame's GPU work fill the stretches in which a frame waits for its host (the serial rANS decode of
every level, the encoder's LIFO pushes)?   usage: int_pipeline_bench.py [steps=24]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from fastpcc_amd import replicas; replicas.bind_to_device_numa_node(0)
from fastpcc_amd.codecs.lossl_coord_int import Config, Model
from fastpcc_amd.codecs.lossl_coord_int.init_random import randomize_
from fastpcc_amd.serving import FramePipeline
from fastpcc_amd.synthetic import batched, lidar_cloud
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 24
model = Model(Config(), 'cuda'); randomize_(model, 1); model = model.cuda().eval()
sweeps = [torch.from_numpy(batched(lidar_cloud(3 + i))).cuda() for i in range(8)]
want = [model.compress(s) for s in sweeps]
for B in (1, 8):
    batch = sweeps[:B]
    n = sum(s.shape[0] for s in batch)

    def step(m, _):
        data = m.compress_many(batch)
        rec = m.decompress_many(data)
        if data != want[:B] or [r.shape[0] for r in rec] != [s.shape[0] for s in batch]:
            raise RuntimeError('a pipelined step did not reproduce the single-sweep streams')
        return None
    for D in (1, 2, 3):
        with FramePipeline(model, depth=D) as pipe:
            pipe.map(step, range(max(D, 2)))
            torch.cuda.synchronize(); t0 = time.perf_counter()
            pipe.map(step, range(steps))
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f'B={B} D={D}: {dt / steps * 1e3:7.2f} ms per step  {n * steps / dt / 1e6:6.2f} Mpoints/s', flush=True)
