#!/usr/bin/env python3
"""Integer codec soak: full-size LiDAR-like sweeps of different seeds and sizes -- lossless round trip, the level-per-call path against the
module-by-module one, batches against single sweeps.  (Level sizes land on both sides of the 8192-row switch between the offset-split and
the tiled convolution from seed to seed.)   usage: int_soak.py [seeds=12]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from fastpcc_amd.codecs.lossl_coord_int import Config, Model, model as M
from fastpcc_amd.codecs.lossl_coord_int.init_random import randomize_
from fastpcc_amd.synthetic import batched, lidar_cloud
seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 12
model = Model(Config(), 'cuda'); randomize_(model, 1); model = model.cuda().eval()
key = lambda a: np.sort(a[:, 0].astype(np.int64) << 32 | a[:, 1].astype(np.int64) << 16 | a[:, 2].astype(np.int64))
streams, frames = [], []
for s in range(seeds):
    beams, az = (64, 2048) if s % 3 == 0 else ((32, 1024) if s % 3 == 1 else (48, 1536))
    xyz = lidar_cloud(100 + s, beams=beams, azimuths=az)
    frame = torch.from_numpy(batched(xyz)).cuda()
    data = model.compress(frame)
    rec = model.decompress(data).cpu().numpy()
    assert (key(rec) == key(xyz)).all(), f'seed {s}: not lossless'
    M.FAST_LEVELS = False
    try:
        assert model.compress(frame) == data, f'seed {s}: the two traversal paths write different bytes'
    finally:
        M.FAST_LEVELS = True
    streams.append(data); frames.append(frame)
    print(f'seed {s}: {len(xyz)} voxels, {len(data)} bytes, lossless, paths agree', flush=True)
for a in range(0, seeds, 4):
    many = model.compress_many(frames[a:a + 4])
    assert many == streams[a:a + 4], f'batch {a}: streams differ from the single sweeps'
    back = model.decompress_many(many)
    assert [b.shape[0] for b in back] == [f.shape[0] for f in frames[a:a + 4]]
print('soak ok')
