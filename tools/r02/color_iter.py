import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from fastpcc_amd import engine as ME
from fastpcc_amd.synthetic import SCALE, batched, body_cloud, enliven
from fastpcc_amd.codecs.lossy_coord_v2 import Model as V2
from fastpcc_amd.codecs.lossy_coord_v2.model_config import baseline_r1
from fastpcc_amd.codecs.lossy_coord_lossy_color import Model as ColorModel
from fastpcc_amd.codecs.lossy_coord_lossy_color.model_config import baseline_r1 as color_cfg
dev = torch.device('cuda:0')
# the headline workload first, as in bench.py
torch.manual_seed(0)
m2 = V2(baseline_r1()); enliven(m2, 0); m2 = m2.to(dev).eval()
f2 = torch.from_numpy(batched(body_cloud(1024, SCALE[1024], seed=2))).to(dev)
for _ in range(6):
    d = m2.compress(f2); ME.clear_global_coordinate_manager(); m2.decompress(d); ME.clear_global_coordinate_manager()
torch.cuda.synchronize()
if len(sys.argv) > 1: torch.cuda.empty_cache()
torch.manual_seed(0)
model = ColorModel(color_cfg()); enliven(model, 3, gain=2.3); model = model.to(dev).eval()
xyz = body_cloud(2048, SCALE.get(2048, 1.0), seed=2)
rng = np.random.default_rng(0)
rgb = torch.from_numpy(np.clip(127 + 90 * np.sin(xyz / 9.0) + rng.normal(0, 8, xyz.shape), 0, 255).astype(np.uint8)).to(dev)
frame = torch.from_numpy(batched(xyz)).to(dev)
for it in range(8):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    data = model.compress(frame, rgb); torch.cuda.synchronize(); t1 = time.perf_counter()
    ME.clear_global_coordinate_manager()
    rec = model.decompress(data); torch.cuda.synchronize(); t2 = time.perf_counter()
    ME.clear_global_coordinate_manager()
    st = torch.cuda.memory_stats()
    print(it, f'enc {1e3*(t1-t0):.1f} dec {1e3*(t2-t1):.1f} ms  reserved {torch.cuda.memory_reserved()/2**30:.1f} GiB  alloc_retries {st["num_alloc_retries"]} segments {st["segment.all.allocated"]}')
