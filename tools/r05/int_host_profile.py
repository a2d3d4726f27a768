#!/usr/bin/env python3
"""Integer codec (cfg#3): where the HOST spends its time between launches (cProfile over 10 x compress / decompress of the LiDAR-like frame)."""
import os, sys, cProfile, pstats, io
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from fastpcc_amd.codecs.lossl_coord_int import Model, Config
from fastpcc_amd.codecs.lossl_coord_int.init_random import randomize_
from fastpcc_amd.synthetic import lidar_cloud, batched
model = Model(Config(), 'cuda'); randomize_(model, 1); model = model.cuda().eval()
frame = torch.from_numpy(batched(lidar_cloud(3))).cuda()
for _ in range(3):
    data = model.compress(frame); rec = model.decompress(data)
torch.cuda.synchronize()
for name, fn in (('compress', lambda: model.compress(frame)), ('decompress', lambda: model.decompress(data))):
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    pr.disable()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(sys.argv[1] if len(sys.argv) > 1 else 'tottime').print_stats(45)
    print(f'## {name} x 10\n' + s.getvalue())
