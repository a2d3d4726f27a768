import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from fastpcc_amd import engine as ME
from fastpcc_amd.synthetic import SCALE, batched, body_cloud, enliven, lidar_cloud
dev = torch.device('cuda:0')
which = sys.argv[1]
if which == 'v2':
    from fastpcc_amd.codecs.lossy_coord_v2 import Model
    from fastpcc_amd.codecs.lossy_coord_v2.model_config import baseline_r1
    torch.manual_seed(0); model = Model(baseline_r1()); enliven(model, 0); model = model.to(dev).eval()
    frame = torch.from_numpy(batched(body_cloud(1024, SCALE[1024], seed=2))).to(dev)
    enc, dec = (lambda: model.compress(frame)), (lambda d: model.decompress(d))
else:
    from fastpcc_amd.codecs.lossl_coord_int import Config, Model
    from fastpcc_amd.codecs.lossl_coord_int.init_random import randomize_
    model = Model(Config(), 'cuda'); randomize_(model, 1); model = model.to(dev).eval()
    frame = torch.from_numpy(batched(lidar_cloud(3))).to(dev)
    enc, dec = (lambda: model.compress(frame)), (lambda d: model.decompress(d))
for it in range(14):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    data = enc(); torch.cuda.synchronize(); t1 = time.perf_counter()
    ME.clear_global_coordinate_manager()
    rec = dec(data); torch.cuda.synchronize(); t2 = time.perf_counter()
    ME.clear_global_coordinate_manager()
    st = torch.cuda.memory_stats()
    print(which, it, f'enc {1e3*(t1-t0):.2f} dec {1e3*(t2-t1):.2f} ms  reserved {torch.cuda.memory_reserved()/2**30:.2f} GiB segments {st["segment.all.allocated"]}')
