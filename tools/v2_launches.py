#!/usr/bin/env python3
"""Launch census of one encode and one decode of the headline workload: kernels by name, device-busy time against the wall clock."""
import collections, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
from torch.profiler import profile, ProfilerActivity
from fastpcc_amd.synthetic import enliven
from fastpcc_amd import engine as ME
from fastpcc_amd.codecs.lossy_coord_v2 import Model
from fastpcc_amd.codecs.lossy_coord_v2.model_config import baseline_r1
from fastpcc_amd.synthetic import SCALE, batched, body_cloud
torch.manual_seed(0)
model = Model(baseline_r1()); enliven(model, 0); model = model.cuda().eval()
frame = torch.from_numpy(batched(body_cloud(1024, SCALE.get(1024, 1.0), seed=2))).cuda()
def enc():
    d = model.compress(frame); torch.cuda.synchronize(); ME.clear_global_coordinate_manager(); return d
def dec(d):
    r = model.decompress(d); torch.cuda.synchronize(); ME.clear_global_coordinate_manager(); return r
for _ in range(3):
    data = enc(); dec(data)
for name, fn in (('encode', enc), ('decode', lambda: dec(data))):
    t0 = time.perf_counter(); fn(); wall = (time.perf_counter() - t0) * 1e3
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        fn()
    agg = collections.defaultdict(lambda: [0, 0.0])
    spans = []
    for e in prof.events():
        if e.device_type == torch.autograd.DeviceType.CUDA:
            agg[e.name[:100]][0] += 1
            agg[e.name[:100]][1] += e.device_time
            spans.append((e.time_range.start, e.time_range.end, e.name[:60]))
    spans.sort()
    # union of the device intervals (streams overlap) and the largest idle gaps
    busy, cur_s, cur_e, gaps, last_name = 0.0, None, None, [], ''
    for s, e, nm in spans:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                busy += cur_e - cur_s; gaps.append((s - cur_e, cur_e - spans[0][0], last_name, nm))
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
        last_name = nm
    busy += cur_e - cur_s
    n = sum(v[0] for v in agg.values())
    print(f'== {name}: wall {wall:.1f} ms (unprofiled), device launches {n}, device busy (union) {busy / 1e3:.1f} ms, '
          f'sum of kernels {sum(v[1] for v in agg.values()) / 1e3:.1f} ms')
    for g, at, before, after in sorted(gaps, reverse=True)[:6]:
        print(f'   idle {g:6.0f} us at {at / 1e3:5.1f} ms: after [{before}] before [{after}]')
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(os.environ.get('TOP', 10))]:
        print(f'{v[0]:5d} {v[1] / 1e3:8.2f} ms  {k}')
