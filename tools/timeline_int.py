#!/usr/bin/env python3
"""Encode / decode wall clock of the integer lossless codec on the LiDAR-like frame (cfg#3)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
from fastpcc_amd import replicas; replicas.bind_to_device_numa_node(0)       # as bench.py does
from fastpcc_amd.codecs.lossl_coord_int import Model, Config
from fastpcc_amd.codecs.lossl_coord_int.init_random import randomize_
from fastpcc_amd.synthetic import lidar_cloud, batched
xyz = lidar_cloud(3)
model = Model(Config(), 'cuda'); randomize_(model, 1); model = model.cuda().eval()
frame = torch.from_numpy(batched(xyz)).cuda()
for it in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    data = model.compress(frame); torch.cuda.synchronize(); t1 = time.perf_counter()
    rec = model.decompress(data); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f'{len(xyz)} voxels: enc {1e3*(t1-t0):.2f} ms dec {1e3*(t2-t1):.2f} ms bytes {len(data)} bpp {8*len(data)/len(xyz):.3f} lossless {rec.shape[0]==len(xyz)}')
