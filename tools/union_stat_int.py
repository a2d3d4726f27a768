#!/usr/bin/env python3
"""Executed / algorithmic MFMA work of the 3x3x3 int8 convolutions on the LiDAR-like frame (cfg#3), per octree level:
32-row blocks in Morton order against neighbour-pattern order."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
from fastpcc_amd import hipops as ops
from fastpcc_amd.int_sparse_conv import _kernel_table
from fastpcc_amd.synthetic import lidar_cloud, batched
xyz = torch.from_numpy(batched(lidar_cloud(3))).cuda()
xyz = xyz - torch.nn.functional.pad(xyz.amin(0)[1:], (1, 0))
_, perm = ops.sort_keys(ops.morton3d_encode(xyz[:, 1:], (2, 1, 0)))
c = xyz[perm.long()].contiguous()
for level in range(8):
    n = c.shape[0]
    _, table = _kernel_table(c, c, (3, 3, 3), (1, 1, 1), None)
    present = (table[:n] > 0).t().contiguous()          # [27, n]
    alg = int(present.sum())
    order = ops.conv_row_order((table - 1).contiguous(), 27, 1, 27, n, 17)
    line = f'level {level} rows {n} pairs/row {alg / n:.2f} | executed/algorithmic (32-row blocks):'
    for name, pm in (('morton', None), ('pattern', order)):
        p = present if pm is None else present[:, pm.long()]
        pad = (-n) % 32
        q = torch.nn.functional.pad(p, (0, pad)).reshape(27, -1, 32).any(2)
        line += f'  {name} {int(q.sum()) * 32 / alg:.2f}'
    print(line)
    c = c.clone(); c[:, 1:] >>= 1
    c = torch.unique_consecutive(c, dim=0)
