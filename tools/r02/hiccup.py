import os, sys, time, gc
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from fastpcc_amd import engine as ME
from fastpcc_amd.synthetic import SCALE, batched, body_cloud, enliven
from fastpcc_amd.codecs.lossy_coord_v2 import Model
from fastpcc_amd.codecs.lossy_coord_v2.model_config import baseline_r1
dev = torch.device('cuda:0')
torch.manual_seed(0); model = Model(baseline_r1()); enliven(model, 0); model = model.to(dev).eval()
frame = torch.from_numpy(batched(body_cloud(1024, SCALE[1024], seed=2))).to(dev)
if len(sys.argv) > 1 and sys.argv[1] == 'nogc':
    gc.disable()
ts = []
for it in range(45):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    data = model.compress(frame); torch.cuda.synchronize(); t1 = time.perf_counter()
    ME.clear_global_coordinate_manager()
    rec = model.decompress(data); torch.cuda.synchronize(); t2 = time.perf_counter()
    ME.clear_global_coordinate_manager()
    if it >= 5: ts.append((1e3 * (t1 - t0), 1e3 * (t2 - t1)))
tot = sorted(a + b for a, b in ts)
print(sys.argv[1:] , 'median %.2f  mean %.2f  p90 %.2f  max %.2f' % (tot[len(tot)//2], sum(tot)/len(tot), tot[int(0.9*len(tot))], tot[-1]), 'gc counts', gc.get_count(), [round(x,1) for x in tot[-6:]])
