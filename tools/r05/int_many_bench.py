#!/usr/bin/env python3
"""Integer codec (cfg#3): B LiDAR-like sweeps through one traversal (compress_many / decompress_many) against one sweep at a time.
usage: int_many_bench.py [reps=5] [Bs=1,2,4,8]"""
import os, sys, time, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from fastpcc_amd import replicas; replicas.bind_to_device_numa_node(0)       # as bench.py does
from fastpcc_amd.codecs.lossl_coord_int import Config, Model
from fastpcc_amd.codecs.lossl_coord_int.init_random import randomize_
from fastpcc_amd.synthetic import batched, lidar_cloud
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
Bs = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else '1,2,4,8').split(',')]
model = Model(Config(), 'cuda'); randomize_(model, 1); model = model.cuda().eval()
frames = [torch.from_numpy(batched(lidar_cloud(3 + i))).cuda() for i in range(max(Bs))]
print('sweeps', [f.shape[0] for f in frames], flush=True)
alone = None
for B in Bs:
    batch = frames[:B]
    te, td = [], []
    for it in range(reps + 2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        data = model.compress_many(batch)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        rec = model.decompress_many(data)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        if it >= 2:
            te.append(t1 - t0); td.append(t2 - t1)
    if B == 1:
        alone = data[0]
    else:
        assert data[0] == alone, 'stream of sweep 0 differs from the one coded alone'
    assert [r.shape[0] for r in rec] == [f.shape[0] for f in batch]
    n = sum(f.shape[0] for f in batch)
    e, d = statistics.median(te) * 1e3, statistics.median(td) * 1e3
    print(f'B={B}: enc {e:.2f} ms dec {d:.2f} ms -> {n / (e + d) / 1e3:.2f} Mpoints/s ({(e + d) / B:.2f} ms per sweep)', flush=True)
