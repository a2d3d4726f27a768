#!/usr/bin/env python3
"""Per-shape table of the weight-gradient launches of one training step."""
import collections, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
from fastpcc_amd import hipops, autograd as AG
from fastpcc_amd.codecs.lossy_coord_v2 import Model
from fastpcc_amd.codecs.lossy_coord_v2.model_config import baseline_r1
from fastpcc_amd.train import TrainConfig, Trainer, synthetic_batches
torch.manual_seed(0)
cfg = TrainConfig()
tr = Trainer(Model(baseline_r1()), cfg, torch.device('cuda', 0))
data = synthetic_batches(0, 1, cfg, torch.device('cuda', 0))
for _ in range(2):
    tr.step(next(data))
trace = []
orig = hipops.conv_wgrad
def traced(x, dy, n, **kw):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); out = orig(x, dy, n, **kw); e1.record()
    nbr = kw.get('nbr')
    trace.append((e0, e1, x.shape[1], dy.shape[1], n, kw.get('n_offsets', 1), kw.get('groups', 1), nbr, kw.get('nbr_ks', 0), kw.get('nbr_os', 1)))
    return out
hipops.conv_wgrad = traced
AG.ops.conv_wgrad = traced
tr.step(next(data))
torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
for e0, e1, ci, co, n, k, g, nbr, ks, os_ in trace:
    if nbr is None:
        pairs = n
    else:
        pairs = int((nbr >= 0).sum()) if nbr.numel() <= 40_000_000 else n * k
    key = (ci, co, n, k, g)
    agg[key][0] += 1; agg[key][1] += e0.elapsed_time(e1); agg[key][2] += 2.0 * pairs * ci * co
tot = sum(v[1] for v in agg.values())
print(f'{len(trace)} wgrad launches, {tot:.2f} ms')
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
    print(k, v[0], f'{v[1]:.3f} ms', f'{v[2] / v[1] / 1e9:.1f} TFLOP/s')
