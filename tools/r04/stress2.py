#!/usr/bin/env python3
"""Race detector for the encoder's host hand-over: DIFFERENT frames alternate, so a coder job that read its pinned buffer before
the copy landed sees the other frame's data (with one frame repeated the stale content equals the fresh one and the race is
invisible).  Every frame's bytes must equal its own reference.  usage: stress2.py [n=300] [resolution=1024]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from fastpcc_amd import engine as ME
from fastpcc_amd.codecs.lossy_coord_v2 import Model
from fastpcc_amd.codecs.lossy_coord_v2.model_config import baseline_r1
from fastpcc_amd.synthetic import SCALE, batched, body_cloud, enliven

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
res = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
torch.manual_seed(0)
model = Model(baseline_r1())
enliven(model, 0)
model = model.cuda().eval()
frames = [torch.from_numpy(batched(body_cloud(res, SCALE.get(res, 1.0) * s, seed=2 + i))).cuda() for i, s in enumerate((1.0, 0.93, 1.0, 0.8))]
print('frames', [f.shape[0] for f in frames], flush=True)
refs = [None] * len(frames)
bad = 0
t0 = time.time()
rng = np.random.default_rng(0)
for i in range(n):
    k = int(rng.integers(len(frames))) if i >= len(frames) else i
    try:
        data = model.compress(frames[k])
    except Exception as e:
        print(f'iteration {i} frame {k}: compress raised {e!r}', flush=True)
        bad += 1
        ME.clear_global_coordinate_manager()
        continue
    ME.clear_global_coordinate_manager()
    if refs[k] is None:
        refs[k] = data
        # the reference itself must be right: it decodes to the frame's size
        rec = model.decompress(data); torch.cuda.synchronize(); ME.clear_global_coordinate_manager()
        print(f'frame {k}: {len(data)} bytes, decoded {rec.shape[0]} of {frames[k].shape[0]}', flush=True)
        continue
    if data != refs[k]:
        bad += 1
        first = next((j for j in range(min(len(data), len(refs[k]))) if data[j] != refs[k][j]), -1)
        print(f'iteration {i} frame {k}: bytes differ ({len(data)} vs {len(refs[k])}, first at {first})', flush=True)
    if i % 5 == 4:
        rec = model.decompress(data); torch.cuda.synchronize(); ME.clear_global_coordinate_manager()
print(f'{n} iterations, {bad} deviations, {time.time() - t0:.1f} s')
