#!/usr/bin/env python3
"""time of the on-device pc_error replacement (D1 + D2 + Hausdorff) on a cfg#2-sized pair: the 997 645-voxel frame against a jittered copy"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from fastpcc_amd.evaluators import d1_metrics, pc_error_metrics
from fastpcc_amd.synthetic import SCALE, body_cloud
xyz = body_cloud(1024, SCALE[1024], seed=2)
rng = np.random.default_rng(0)
rec = np.unique(np.clip(xyz + rng.integers(-1, 2, xyz.shape), 0, 1023), axis=0)
a, b = torch.from_numpy(xyz).cuda(), torch.from_numpy(rec).cuda()
for name, fn in (('d1_metrics', lambda: d1_metrics(a, b, 1024)), ('pc_error_metrics (D1 + D2 + Hausdorff, PCA normals over 30 neighbours)', lambda: pc_error_metrics(a, b, 1024, hausdorff=True))):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter(); out = fn(); torch.cuda.synchronize()
    print(f'{name}: {(time.perf_counter() - t0) * 1e3:.1f} ms for {len(xyz)} vs {len(rec)} voxels')
print({k: round(v, 4) for k, v in out.items() if 'PSNR' in k})
