#!/usr/bin/env python3
"""Determinism stress: the same frame compressed N times in one process must give the same bytes every time; on a deviation the
symbols kept by the coder tell which level went wrong.  usage: stress.py [n=150] [resolution=1024]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from fastpcc_amd import engine as ME, hipops
from fastpcc_amd.codecs.lossy_coord_v2 import Model
from fastpcc_amd.codecs.lossy_coord_v2.model_config import baseline_r1
from fastpcc_amd.synthetic import SCALE, batched, body_cloud, enliven

n = int(sys.argv[1]) if len(sys.argv) > 1 else 150
res = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
torch.manual_seed(0)
model = Model(baseline_r1())
enliven(model, 0)
model = model.cuda().eval()
model.em_lossless_based.keep_symbols = True
frame = torch.from_numpy(batched(body_cloud(res, SCALE.get(res, 1.0), seed=2))).cuda()
ref = ref_sym = None
bad = 0
t0 = time.time()
for i in range(n):
    try:
        data = model.compress(frame)
    except Exception as e:
        print(f'iteration {i}: compress raised {e!r}', flush=True)
        bad += 1
        ME.clear_global_coordinate_manager()
        continue
    sym = model.em_lossless_based.last_symbols
    ME.clear_global_coordinate_manager()
    if ref is None:
        ref, ref_sym = data, sym
        continue
    if data != ref:
        bad += 1
        msg = f'iteration {i}: bytes differ ({len(data)} vs {len(ref)})'
        for k in ('residual', 'occupancy', 'prob'):
            a, b = np.asarray(sym[k]).reshape(-1), np.asarray(ref_sym[k]).reshape(-1)
            if a.shape != b.shape:
                msg += f'; {k}: shape {a.shape} vs {b.shape}'
            else:
                d = np.nonzero(a != b)[0]
                if d.size:
                    sizes = np.cumsum([0] + list(ref_sym['sizes']))
                    msg += f'; {k}: {d.size} differ, first at {d[0]} (level boundaries {sizes.tolist() if k != "residual" else ""})'
        print(msg, flush=True)
    if i % 3 == 2:       # decode too now and then (its launches change what the allocator hands out)
        rec = model.decompress(data)
        torch.cuda.synchronize()
        ME.clear_global_coordinate_manager()
print(f'{n} iterations, {bad} deviations, {time.time() - t0:.1f} s')
