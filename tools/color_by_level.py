#!/usr/bin/env python3
"""Every convolution-family launch of one encode and one decode of the colour codec (cfg#4, 2 M coloured voxels at 2048^3) with HIP
events around it, in the format of `bench.py --dump-trace` (roll it up with profiles/conv_by_level.py).
usage: color_by_level.py <out.txt>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import torch
import bench
from fastpcc_amd import engine as ME, hipops
from fastpcc_amd.codecs.lossy_coord_lossy_color import Model
from fastpcc_amd.codecs.lossy_coord_lossy_color.model_config import baseline_r1
from fastpcc_amd.synthetic import SCALE, batched, body_cloud, enliven
res = 2048
torch.manual_seed(0)
model = Model(baseline_r1()); enliven(model, 0); model = model.cuda().eval()
xyz = body_cloud(res, SCALE.get(res, 1.0), seed=4)
rng = np.random.default_rng(4)
u = xyz / float(res)
col = 127.5 + 100 * np.stack([np.sin(9 * u[:, 0] + 2 * u[:, 1]), np.cos(7 * u[:, 1] - 3 * u[:, 2]), np.sin(5 * u[:, 2] + u[:, 0])], 1)
col = np.clip(col + rng.normal(0, 8, col.shape), 0, 255).astype(np.float32)
frame = torch.from_numpy(batched(xyz)).cuda()
color = torch.from_numpy(col).cuda()
for _ in range(3):
    data = model.compress(frame, color); ME.clear_global_coordinate_manager()
    model.decompress(data); ME.clear_global_coordinate_manager()
torch.cuda.synchronize()
hipops.reserve_trace_events(2000)
with open(sys.argv[1], 'w') as f:
    f.write('kind c_in c_out n_out n_off groups ms algo_gflop algo_tflops dense_tflops\n')
    for name, fn in (('encode', lambda: model.compress(frame, color)), ('decode', lambda: model.decompress(data))):
        hipops.CONV_TRACE = []
        fn()
        torch.cuda.synchronize()
        trace, hipops.CONV_TRACE = hipops.CONV_TRACE, None
        cache = {}
        tot = fl_tot = 0.0
        shapes = {}
        for ev0, ev1, info in trace:
            dt = ev0.elapsed_time(ev1)
            fl = bench.conv_flops(info, cache)
            dense = 2.0 * info['n_out'] * info['groups'] * info['n_offsets'] * info['c_in'] * info['c_out']
            f.write(f"{'mfma' if info['mfma'] else 'valu'} {info['c_in']} {info['c_out']} {info['n_out']} {info['n_offsets']} {info['groups']} "
                    f"{dt:.4f} {fl / 1e9:.3f} {fl / dt / 1e9:.2f} {dense / dt / 1e9:.2f}\n")
            tot += dt; fl_tot += fl
            pairs = fl / (2.0 * info['c_in'] * info['c_out'])
            e = shapes.setdefault((info['n_out'], info['n_offsets'], info['c_in'], info['c_out'], 'mfma' if info['mfma'] else 'valu'), [0, 0.0, 0.0, 0.0])
            e[0] += 1; e[1] += dt; e[2] += fl; e[3] += 4.0 * (pairs * info['c_in'] + info['n_out'] * info['groups'] * info['c_out'])
        print(f'## {name}: {len(trace)} convolution-family launches, {tot:.2f} ms, {fl_tot / 1e9:.0f} algorithmic GFLOP, {fl_tot / tot / 1e9:.1f} TFLOP/s')
        print()
        print('| rows | offsets | C_in -> C_out | kernel family | launches | ms | TFLOP/s | % of 157.3 | gathered + written GB/s (algorithmic) | % of 8 TB/s |')
        print('|---:|---:|---|---|---:|---:|---:|---:|---:|---:|')
        for (n_out, n_off, ci, co, kind), (cnt, ms, fl, by) in sorted(shapes.items(), key=lambda kv: -kv[1][1])[:14]:
            print(f'| {n_out} | {n_off} | {ci} -> {co} | {kind} | {cnt} | {ms:.3f} | {fl / ms / 1e9:.1f} | {fl / ms / 1e9 / 157.3 * 100:.1f} | {by / ms / 1e6:.0f} | {by / ms / 1e6 / 8000 * 100:.1f} |')
        print()
        ME.clear_global_coordinate_manager()
