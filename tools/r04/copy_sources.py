#!/usr/bin/env python3
"""Where the ~130 copies per frame of the v2 codec come from: aten::copy_ / aten::to / aten::contiguous / aten::cat calls of one encode +
decode by Python source line (torch profiler with stacks)."""
import collections, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from torch.profiler import profile, ProfilerActivity
from fastpcc_amd import engine as ME
from fastpcc_amd.codecs.lossy_coord_v2 import Model
from fastpcc_amd.codecs.lossy_coord_v2.model_config import baseline_r1
from fastpcc_amd.synthetic import SCALE, batched, body_cloud, enliven
torch.manual_seed(0)
model = Model(baseline_r1()); enliven(model, 0); model = model.cuda().eval()
frame = torch.from_numpy(batched(body_cloud(1024, SCALE[1024], seed=2))).cuda()
def step():
    data = model.compress(frame); torch.cuda.synchronize(); ME.clear_global_coordinate_manager()
    rec = model.decompress(data); torch.cuda.synchronize(); ME.clear_global_coordinate_manager()
for _ in range(3): step()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
agg = collections.Counter()
names = collections.Counter()
for e in prof.events():
    if e.name in ('aten::copy_', 'aten::_to_copy', 'aten::contiguous', 'aten::cat', 'aten::clone', 'aten::index_select', 'aten::fill_', 'aten::zero_'):
        names[e.name] += 1
        st = [s for s in (e.stack or []) if '/fastpcc_amd/' in s]
        agg[(e.name, st[0].split('/fastpcc_amd/')[-1] if st else '?')] += 1
print(dict(names))
for (name, where), c in agg.most_common(40):
    print(f'{c:4d}  {name:18s} {where}')
