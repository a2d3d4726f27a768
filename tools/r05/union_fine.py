#!/usr/bin/env python3
"""Executed / algorithmic MFMA work of a 3x3x3 convolution on the FINE maps (where the narrow layers of the colour decoder and the
16 -> 32 layer of cfg#2 run): a block of 32 rows executes every kernel offset any of its rows has.  Natural (Morton) order against the
neighbour-pattern order, and -- what an LDS brick of Morton-contiguous rows would stage -- the distinct input rows per block of
T consecutive Morton rows relative to T (the halo factor).  usage: union_fine.py [resolution=1024]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from fastpcc_amd import engine as ME
from fastpcc_amd.synthetic import SCALE, batched, body_cloud
res = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
frame = torch.from_numpy(batched(body_cloud(res, SCALE.get(res, 1.0), seed=2))).cuda()
cm = ME.CoordinateManager(D=3)
x = ME.SparseTensor(torch.ones((frame.shape[0], 1), device='cuda'), coordinates=frame, coordinate_manager=cm)
m = cm._map(x.coordinate_map_key)
for level in range(0, 2):
    if level:
        m = cm._ensure_parent(m)
    nbr = cm._nbr27(m)                      # [27, n]
    n = m.n
    present = nbr >= 0
    alg = int(present.sum())
    order = ME.ops.conv_row_order(nbr, 27, n, 1, n, cm.ROW_ORDER_WINDOW_LOG2)
    for name, perm in (('natural', None), ('pattern', order)):
        p = present if perm is None else present[:, perm.long()]
        pad = (-n) % 32
        q = torch.nn.functional.pad(p, (0, pad)).reshape(27, -1, 32).any(2)
        print(f'level {level} rows {n} pairs/row {alg / n:.2f} {name}: executed/algorithmic at 32-row blocks {int(q.sum()) * 32 / alg:.3f}')
    for T in (128, 256, 512):
        blocks = n // T
        t = nbr[:, :blocks * T].reshape(27, blocks, T).permute(1, 0, 2).reshape(blocks, 27 * T)
        s, _ = torch.sort(t, dim=1)
        distinct = ((s[:, 1:] != s[:, :-1]) & (s[:, 1:] >= 0)).sum(1) + (s[:, 0] >= 0).long()
        print(f'level {level}: distinct input rows per {T} consecutive Morton rows / {T} = {float(distinct.float().mean()) / T:.3f}')
