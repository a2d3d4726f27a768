#!/usr/bin/env python3
"""Launch census of one encode and one decode of the colour codec (cfg#4): kernels by device time, device-busy time against the wall clock."""
import collections, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import torch
from torch.profiler import profile, ProfilerActivity
from fastpcc_amd.synthetic import enliven
from fastpcc_amd import engine as ME
from fastpcc_amd.codecs.lossy_coord_lossy_color import Model
from fastpcc_amd.codecs.lossy_coord_lossy_color.model_config import baseline_r1
from fastpcc_amd.synthetic import SCALE, batched, body_cloud
res = 2048
torch.manual_seed(0)
model = Model(baseline_r1()); enliven(model, 0); model = model.cuda().eval()
xyz = body_cloud(res, SCALE.get(res, 1.0), seed=4)
u = xyz / float(res)
col = np.clip(127.5 + 100 * np.stack([np.sin(9 * u[:, 0]), np.cos(7 * u[:, 1]), np.sin(5 * u[:, 2])], 1), 0, 255).astype(np.float32)
frame = torch.from_numpy(batched(xyz)).cuda(); color = torch.from_numpy(col).cuda()
def enc():
    d = model.compress(frame, color); torch.cuda.synchronize(); ME.clear_global_coordinate_manager(); return d
def dec(d):
    r = model.decompress(d); torch.cuda.synchronize(); ME.clear_global_coordinate_manager(); return r
for _ in range(3):
    data = enc(); dec(data)
for name, fn in (('encode', enc), ('decode', lambda: dec(data))):
    t0 = time.perf_counter(); fn(); wall = (time.perf_counter() - t0) * 1e3
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        fn()
    agg = collections.defaultdict(lambda: [0, 0.0])
    for e in prof.events():
        if e.device_type == torch.autograd.DeviceType.CUDA:
            agg[e.name[:100]][0] += 1; agg[e.name[:100]][1] += e.device_time
    print(f'== {name}: wall {wall:.1f} ms, launches {sum(v[0] for v in agg.values())}, sum of kernels {sum(v[1] for v in agg.values()) / 1e3:.1f} ms')
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:12]:
        print(f'{v[0]:5d} {v[1] / 1e3:8.2f} ms  {k}')

# per-shape table of the decode's convolutions
from fastpcc_amd import hipops
hipops.CONV_TRACE = []
dec(data)
trace, hipops.CONV_TRACE = hipops.CONV_TRACE, None
agg = collections.defaultdict(lambda: [0, 0.0])
for e0, e1, info in trace:
    key = ('mfma' if info['mfma'] else 'valu', info['c_in'], info['c_out'], info['n_out'], info['n_offsets'], info['groups'])
    agg[key][0] += 1; agg[key][1] += e0.elapsed_time(e1)
print('== decode convolutions by shape')
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
    print(k, v[0], f'{v[1]:.2f} ms')
