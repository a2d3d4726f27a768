#!/usr/bin/env python3
"""Weight gradient of the 3x3x3 layers: active (32-row block, offset) pairs per offset on the maps of a training batch -- how unevenly
the work of the (row split, offset) workgroups of k_wgrad_rows is spread over the 27 offsets, and how full the executed blocks are."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from fastpcc_amd import engine as ME
from fastpcc_amd.train import TrainConfig, synthetic_batches
cfg = TrainConfig()
batch = next(synthetic_batches(0, 1, cfg, torch.device('cuda', 0), 128))
cm = ME.CoordinateManager(D=3)
t = ME.SparseTensor(torch.ones((batch.xyz.shape[0], 1), device='cuda'), coordinates=batch.xyz.int().cuda(), coordinate_manager=cm)
cm.build_pyramid(t.coordinate_map_key, 4)
m = cm._map(t.coordinate_map_key)
while m is not None and m.n > 200:
    nbr = cm._nbr27(m)                                    # [27, n]
    order = cm._row_order(m, training=True)
    rows = nbr if order is None or order is False else nbr[:, order.long()]
    n = m.n
    present = torch.nn.functional.pad(rows >= 0, (0, (-n) % 32)).view(27, -1, 32)
    active = present.any(2).sum(1).tolist()              # executed blocks per offset
    pairs = (rows >= 0).sum(1).tolist()
    print(f'rows {n:7d} blocks {(n + 31) // 32:5d}: executed blocks per offset min {min(active)} median {sorted(active)[13]} max {max(active)} '
          f'(mean {sum(active) / 27:.0f}); max / mean = {max(active) * 27 / sum(active):.2f}; rows with the offset inside executed blocks '
          f'{sum(pairs) / (32 * sum(active)):.2f}')
    m = m.parent
