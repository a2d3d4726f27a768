#!/usr/bin/env python3
"""k_nbr27_from_parent on the levels of the cfg#2 frame: time and written bytes per launch (108 B per row; 4 B for the mask form)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from fastpcc_amd import engine as ME, hipops as ops
from fastpcc_amd.synthetic import SCALE, batched, body_cloud
frame = torch.from_numpy(batched(body_cloud(1024, SCALE[1024], seed=2))).cuda()
cm = ME.CoordinateManager(D=3)
x = ME.SparseTensor(torch.ones((frame.shape[0], 1), device='cuda'), coordinates=frame, coordinate_manager=cm)
m0 = cm._map(x.coordinate_map_key)
cm.build_pyramid(x.coordinate_map_key, 6)
def timed(fn, reps=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e6
m = m0
while m.parent is not None and m.parent.n > 2048:
    pn = cm._nbr27(m.parent)
    t = timed(lambda: ops.nbr27_from_parent(m.keys, m.parent_of, pn, m.child_row))
    print(f'table  {m.n:8d} rows from {m.parent.n:7d} parents: {t:7.1f} us  {108 * m.n / t / 1e6:5.2f} TB/s written')
    if m is m0:
        t = timed(lambda: ops.mask27_from_parent(m.keys, m.parent_of, pn, m.child_row))
        print(f'masks  {m.n:8d} rows from {m.parent.n:7d} parents: {t:7.1f} us')
    g = cm._generated(m.parent)
    t = timed(lambda: ops.nbr27_from_parent(None, None, pn, None, n=g.n))
    print(f'gen    {g.n:8d} rows from {m.parent.n:7d} parents: {t:7.1f} us  {108 * g.n / t / 1e6:5.2f} TB/s written')
    m = m.parent
