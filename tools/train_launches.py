#!/usr/bin/env python3
"""Launch census of one training step: kernels by name (count, device time), device-busy total against the wall clock."""
import collections, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from torch.profiler import profile, ProfilerActivity
from fastpcc_amd.codecs.lossy_coord_v2 import Model
from fastpcc_amd.codecs.lossy_coord_v2.model_config import baseline_r1
from fastpcc_amd.train import TrainConfig, Trainer, synthetic_batches
torch.manual_seed(0)
cfg = TrainConfig()
tr = Trainer(Model(baseline_r1()), cfg, torch.device('cuda', 0))
data = synthetic_batches(0, 1, cfg, torch.device('cuda', 0))
for _ in range(3):
    tr.step(next(data))
batch = next(data)
torch.cuda.synchronize()
t0 = time.perf_counter()
tr.step(batch)
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) * 1e3
batch = next(data)
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    tr.step(batch)
    torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for e in prof.events():
    if e.device_type == torch.autograd.DeviceType.CUDA:
        agg[e.name[:90]][0] += 1
        agg[e.name[:90]][1] += e.device_time
n = sum(v[0] for v in agg.values()); busy = sum(v[1] for v in agg.values()) / 1e3
print(f'wall {wall:.1f} ms (unprofiled), device launches {n}, device busy {busy:.1f} ms')
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1 if os.environ.get("BY_TIME") else 0])[:int(os.environ.get('TOP', 40))]:
    print(f'{v[0]:5d} {v[1] / 1e3:8.2f} ms  {k}')
if os.environ.get('CPU_OPS'):
    ops = collections.Counter(e.name for e in prof.events() if e.device_type == torch.autograd.DeviceType.CPU and e.name.startswith('aten::'))
    for k, v in ops.most_common(40):
        print(f'{v:5d} {k}')
