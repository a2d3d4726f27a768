#!/usr/bin/env python3
"""Frames in flight at bench scale: four DIFFERENT ~1 M-voxel frames go through the two-context pipeline in random order, with the
stages named as in bench.py; every frame's bytes and decoded cloud must equal what the single-frame code gave.
usage: stress3.py [n=200] [resolution=1024]"""
import os, sys, time, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from fastpcc_amd import engine as ME
from fastpcc_amd.codecs.lossy_coord_v2 import Model
from fastpcc_amd.codecs.lossy_coord_v2.model_config import baseline_r1
from fastpcc_amd.serving import FramePipeline, wait_for_my_work
from fastpcc_amd.synthetic import SCALE, batched, body_cloud, enliven

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
res = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
torch.manual_seed(0)
model = Model(baseline_r1())
enliven(model, 0)
model = model.cuda().eval()
frames = [torch.from_numpy(batched(body_cloud(res, SCALE.get(res, 1.0) * s, seed=2 + i))).cuda() for i, s in enumerate((1.0, 0.93, 1.0, 0.8))]


def digest(points):
    p = points.cpu().numpy().astype(np.int64)
    return hashlib.sha256(np.sort((p[:, 0] << 42) | (p[:, 1] << 21) | p[:, 2]).tobytes()).hexdigest()


refs = []
for f in frames:
    data = model.compress(f); ME.clear_global_coordinate_manager()
    rec = model.decompress(data); torch.cuda.synchronize(); ME.clear_global_coordinate_manager()
    refs.append((data, digest(rec)))
print('frames', [f.shape[0] for f in frames], 'bytes', [len(r[0]) for r in refs], flush=True)
order = np.random.default_rng(1).integers(len(frames), size=n).tolist()
with FramePipeline(model, depth=2) as pipe:
    def step(ctx, k):
        with pipe.stage('compress'):
            data = ctx.compress(frames[k]); ME.clear_global_coordinate_manager()
        with pipe.stage('decompress'):
            rec = ctx.decompress(data); wait_for_my_work(rec.device); ME.clear_global_coordinate_manager()
        return k, data, digest(rec)
    t0 = time.time()
    out = pipe.map(step, order)
    dt = time.time() - t0
bad = sum(1 for k, data, dg in out if data != refs[k][0] or dg != refs[k][1])
pts = sum(frames[k].shape[0] for k in order)
print(f'{n} frames through the pipeline in {dt:.1f} s ({pts / dt / 1e6:.1f} Mpoints/s incl. the digests), {bad} deviations')
sys.exit(1 if bad else 0)
