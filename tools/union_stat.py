#!/usr/bin/env python3
"""Executed / algorithmic MFMA work of a 3x3x3 convolution per pyramid level: a block of B rows executes every kernel offset any of
its rows has; natural (Morton) row order against the neighbour-pattern order, B = 64 / 32 / 16 / 8."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
from fastpcc_amd import engine as ME
from fastpcc_amd.synthetic import SCALE, batched, body_cloud
frame = torch.from_numpy(batched(body_cloud(1024, SCALE.get(1024, 1.0), seed=2))).cuda()
cm = ME.CoordinateManager(D=3)
x = ME.SparseTensor(torch.ones((frame.shape[0], 1), device='cuda'), coordinates=frame, coordinate_manager=cm)
m = cm._map(x.coordinate_map_key)
for level in range(1, 5):
    m = cm._ensure_parent(m)
    nbr = cm._nbr27(m)                      # [27, n]
    order = cm._row_order(m)
    present = (nbr >= 0)                    # [27, n]
    n = m.n
    alg = int(present.sum())
    for name, perm in (('natural', None), ('pattern', order)):
        p = present if perm is None else present[:, perm.long()]
        line = f'level {level} rows {n} {name}: pairs/row {alg / n:.2f} | executed/algorithmic'
        for bs in (64, 32, 16, 8):
            pad = (-n) % bs
            q = torch.nn.functional.pad(p, (0, pad)).reshape(27, -1, bs).any(2)       # [27, blocks]
            executed = int(q.sum()) * bs
            line += f'  {bs}-row {executed / alg:.3f}'
        print(line)
