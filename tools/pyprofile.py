#!/usr/bin/env python3
"""cProfile of the host side of one encode + decode (where the Python time per launch goes)."""
import cProfile, os, pstats, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
from fastpcc_amd.synthetic import enliven
from fastpcc_amd import engine as ME
from fastpcc_amd.codecs.lossy_coord_v2 import Model
from fastpcc_amd.codecs.lossy_coord_v2.model_config import baseline_r1
from fastpcc_amd.synthetic import SCALE, batched, body_cloud
res = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
torch.manual_seed(0)
model = Model(baseline_r1()); enliven(model, 0); model = model.cuda().eval()
frame = torch.from_numpy(batched(body_cloud(res, SCALE.get(res, 1.0), seed=2))).cuda()
def step():
    data = model.compress(frame); torch.cuda.synchronize(); ME.clear_global_coordinate_manager()
    rec = model.decompress(data); torch.cuda.synchronize(); ME.clear_global_coordinate_manager()
for _ in range(3): step()
pr = cProfile.Profile(); pr.enable()
for _ in range(5): step()
pr.disable()
st = pstats.Stats(pr); st.sort_stats('tottime').print_stats(35)
print('---- callers of the blocking calls ----')
st.print_callers('item|synchronize|tolist')
