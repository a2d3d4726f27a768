#!/usr/bin/env python3
"""Launch census of one encode and one decode of the integer codec (cfg#3): kernels by name, device-busy time against the wall clock."""
import collections, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
from fastpcc_amd import replicas; replicas.bind_to_device_numa_node(0)       # as bench.py does
from torch.profiler import profile, ProfilerActivity
from fastpcc_amd.codecs.lossl_coord_int import Model, Config
from fastpcc_amd.codecs.lossl_coord_int.init_random import randomize_
from fastpcc_amd.synthetic import lidar_cloud, batched
xyz = lidar_cloud(3)
model = Model(Config(), 'cuda'); randomize_(model, 1); model = model.cuda().eval()
frame = torch.from_numpy(batched(xyz)).cuda()
for _ in range(3):
    data = model.compress(frame); model.decompress(data)
torch.cuda.synchronize()
for name, fn in (('encode', lambda: model.compress(frame)), ('decode', lambda: model.decompress(data))):
    t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); wall = (time.perf_counter() - t0) * 1e3
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        fn(); torch.cuda.synchronize()
    agg = collections.defaultdict(lambda: [0, 0.0])
    for e in prof.events():
        if e.device_type == torch.autograd.DeviceType.CUDA:
            agg[e.name[:100]][0] += 1
            agg[e.name[:100]][1] += e.device_time
    n = sum(v[0] for v in agg.values()); busy = sum(v[1] for v in agg.values()) / 1e3
    print(f'== {name}: wall {wall:.1f} ms (unprofiled), device launches {n}, device busy {busy:.1f} ms')
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(os.environ.get('TOP', 14))]:
        print(f'{v[0]:5d} {v[1] / 1e3:8.2f} ms  {k}')
