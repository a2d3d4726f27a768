#!/usr/bin/env python3
"""Per-launch table of the convolution calls of one training step (forward + backward), slowest shapes first."""
import collections, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
from fastpcc_amd import hipops
from fastpcc_amd.codecs.lossy_coord_v2 import Model
from fastpcc_amd.codecs.lossy_coord_v2.model_config import baseline_r1
from fastpcc_amd.train import TrainConfig, Trainer, synthetic_batches
torch.manual_seed(0)
cfg = TrainConfig()
tr = Trainer(Model(baseline_r1()), cfg, torch.device('cuda', 0))
data = synthetic_batches(0, 1, cfg, torch.device('cuda', 0))
for _ in range(2):
    tr.step(next(data))
hipops.CONV_TRACE = []
tr.step(next(data))
torch.cuda.synchronize()
trace, hipops.CONV_TRACE = hipops.CONV_TRACE, None
agg = collections.defaultdict(lambda: [0, 0.0])
for e0, e1, info in trace:
    key = ('mfma' if info['mfma'] else 'VALU', info['c_in'], info['c_out'], info['n_out'], info['n_offsets'], info['groups'])
    agg[key][0] += 1
    agg[key][1] += e0.elapsed_time(e1)
tot = sum(v[1] for v in agg.values())
print(f'conv_f32 launches {len(trace)}, total {tot:.1f} ms')
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
    print(k, v[0], f'{v[1]:.2f} ms')
