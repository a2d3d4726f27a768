#!/usr/bin/env python3
"""B independent ~1M-voxel frames through one traversal (compress_many / decompress_many), each half closed by a device synchronise:
Mpoints/s against B, beside the one-frame loop.  usage: many_bench.py [reps=8] [Bs=1,2,3,4] [resolution=1024]"""
import os, sys, time, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from fastpcc_amd import engine as ME
from fastpcc_amd.codecs.lossy_coord_v2 import Model
from fastpcc_amd.codecs.lossy_coord_v2.model_config import baseline_r1
from fastpcc_amd.synthetic import SCALE, batched, body_cloud, enliven

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
Bs = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else '1,2,3,4').split(',')]
res = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
torch.manual_seed(0)
model = Model(baseline_r1())
enliven(model, 0)
model = model.cuda().eval()
frames = [torch.from_numpy(batched(body_cloud(res, SCALE.get(res, 1.0), seed=2 + i))).cuda() for i in range(max(Bs))]
print('frames', [f.shape[0] for f in frames], flush=True)
alone = None
for B in Bs:
    batch = frames[:B]
    te, td = [], []
    for it in range(reps + 3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        data = model.compress_many(batch)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        ME.clear_global_coordinate_manager()
        rec = model.decompress_many(data)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        ME.clear_global_coordinate_manager()
        if it >= 3:
            te.append(t1 - t0); td.append(t2 - t1)
    if B == 1:
        alone = data[0]
    else:
        assert data[0] == alone, 'stream of frame 0 differs from the one coded alone'
    n = sum(f.shape[0] for f in batch)
    e, d = statistics.median(te) * 1e3, statistics.median(td) * 1e3
    print(f'B={B}: enc {e:.2f} ms dec {d:.2f} ms -> {n / (e + d) / 1e3:.2f} Mpoints/s  (decoded {[r.shape[0] for r in rec]})', flush=True)
