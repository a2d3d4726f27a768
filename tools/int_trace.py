#!/usr/bin/env python3
"""Per-shape table of the int8 convolution / linear launches of one encode of the integer codec (cfg#3)."""
import collections, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
from fastpcc_amd import hipops, int_sparse_conv
from fastpcc_amd.codecs.lossl_coord_int import Model, Config
from fastpcc_amd.codecs.lossl_coord_int.init_random import randomize_
from fastpcc_amd.synthetic import lidar_cloud, batched
xyz = lidar_cloud(3)
model = Model(Config(), 'cuda'); randomize_(model, 1); model = model.cuda().eval()
frame = torch.from_numpy(batched(xyz)).cuda()
for _ in range(3):
    data = model.compress(frame)
trace = []
orig = hipops.conv_i8
def traced(a, w, c_in, c_out, n_out, **kw):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); out = orig(a, w, c_in, c_out, n_out, **kw); e1.record()
    nbr = kw.get('nbr')
    pairs = n_out if nbr is None else None
    trace.append((e0, e1, c_in, c_out, n_out, kw.get('n_offsets', 1), nbr, kw.get('out_bits', 32)))
    return out
hipops.conv_i8 = traced; int_sparse_conv.ops.conv_i8 = traced
data = model.compress(frame)
torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
for e0, e1, ci, co, n, k, nbr, bits in trace:
    pairs = n if nbr is None else int((nbr[:n] > 0).sum())
    key = (ci, co, n, k, bits)
    agg[key][0] += 1; agg[key][1] += e0.elapsed_time(e1); agg[key][2] += 2.0 * pairs * ci * co
tot = sum(v[1] for v in agg.values())
print(f'{len(trace)} launches, {tot:.2f} ms, {sum(v[2] for v in agg.values()) / 1e9:.1f} Gop')
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:30]:
    print(k, v[0], f'{v[1]:.3f} ms', f'{v[2] / v[1] / 1e9:.1f} Top/s', f'pairs/row {v[2] / v[0] / (2.0 * k[0] * k[1]) / k[2]:.2f}')
