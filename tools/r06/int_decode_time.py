#!/usr/bin/env python3
"""cfg#3 (lossl_coord_int, 113 K-voxel LiDAR-like sweep): encode / decode wall time, median of 9 after 3 warm-ups.  Environment switches
of the host decoder (FPCC_HOST_WARMERS=0..8) are read by libfpcc_host at first use."""
import os, statistics, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from fastpcc_amd import engine as ME, replicas
from fastpcc_amd.synthetic import batched, lidar_cloud
from fastpcc_amd.codecs.lossl_coord_int import Config, Model
from fastpcc_amd.codecs.lossl_coord_int.init_random import randomize_
torch.cuda.set_device(0)
replicas.bind_to_device_numa_node(0)
dev = torch.device('cuda', 0)
model = Model(Config(), 'cuda'); randomize_(model, 1); model = model.to(dev).eval()
frame = torch.from_numpy(batched(lidar_cloud(3))).to(dev)
te, td = [], []
for it in range(12):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    data = model.compress(frame); torch.cuda.synchronize(); t1 = time.perf_counter()
    ME.clear_global_coordinate_manager()
    rec = model.decompress(data); torch.cuda.synchronize(); t2 = time.perf_counter()
    ME.clear_global_coordinate_manager()
    if it >= 3:
        te.append(t1 - t0); td.append(t2 - t1)
print(f"FPCC_HOST_WARMERS={os.environ.get('FPCC_HOST_WARMERS', 'default')}: encode {statistics.median(te) * 1e3:.2f} ms, decode {statistics.median(td) * 1e3:.2f} ms "
      f"(min {min(td) * 1e3:.2f}), lossless {rec.shape[0] == frame.shape[0]}")
