#!/usr/bin/env python3
"""Encode / decode wall clock of the joint geometry + colour codec on the ~2M-voxel body-surface frame at 2048^3 (cfg#4)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import torch
from fastpcc_amd import replicas; replicas.bind_to_device_numa_node(0)       # as bench.py does
from fastpcc_amd.synthetic import enliven
from fastpcc_amd import engine as ME
from fastpcc_amd.codecs.lossy_coord_lossy_color import Model
from fastpcc_amd.codecs.lossy_coord_lossy_color.model_config import baseline_r1
from fastpcc_amd.evaluators import d1_metrics
from fastpcc_amd.synthetic import SCALE, batched, body_cloud
res = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
torch.manual_seed(0)
model = Model(baseline_r1()); enliven(model, 0); model = model.cuda().eval()
xyz = body_cloud(res, SCALE.get(res, 1.0), seed=4)
rng = np.random.default_rng(4)
u = xyz / float(res)
col = 127.5 + 100 * np.stack([np.sin(9 * u[:, 0] + 2 * u[:, 1]), np.cos(7 * u[:, 1] - 3 * u[:, 2]), np.sin(5 * u[:, 2] + u[:, 0])], 1)
col = np.clip(col + rng.normal(0, 8, col.shape), 0, 255).astype(np.float32)
frame = torch.from_numpy(batched(xyz)).cuda()
color = torch.from_numpy(col).cuda()
for it in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    data = model.compress(frame, color); torch.cuda.synchronize(); t1 = time.perf_counter()
    ME.clear_global_coordinate_manager()
    rec_xyz, rec_col = model.decompress(data); torch.cuda.synchronize(); t2 = time.perf_counter()
    ME.clear_global_coordinate_manager()
    print(f'{len(xyz)} voxels: enc {1e3*(t1-t0):.1f} ms dec {1e3*(t2-t1):.1f} ms -> {len(xyz)/(t2-t0)/1e6:.2f} Mpoints/s, {len(data)} B, '
          f'bpp {8*len(data)/len(xyz):.3f}, decoded {rec_xyz.shape[0]} points', flush=True)
q = d1_metrics(frame[:, 1:], rec_xyz, res, color, rec_col)
print('D1 PSNR %.2f dB, Y PSNR %.2f dB' % (q['mseF,PSNR (p2point)'], q['c[0],PSNRF']))
