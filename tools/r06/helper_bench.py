#!/usr/bin/env python3
"""The bandwidth-bound helper kernels alone, on the maps of a batch of B cfg#2 frames (the sizes bench.py's launches have): time per
launch and ALGORITHMIC bytes / time (not PMC traffic: that is profiles/hbm_bandwidth.py on a rocprofv3 --pmc run).
    python tools/r06/helper_bench.py [B]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from fastpcc_amd import engine as ME, hipops as ops
from fastpcc_amd.synthetic import SCALE, batched, body_cloud

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
clouds = []
for b in range(B):
    c = batched(body_cloud(1024, SCALE[1024], seed=2 + b))
    c[:, 0] = b
    clouds.append(c)
frame = torch.from_numpy(np.concatenate(clouds)).cuda()
cm = ME.CoordinateManager(D=3)
x = ME.SparseTensor(torch.ones((frame.shape[0], 1), device='cuda'), coordinates=frame, coordinate_manager=cm)
m0 = cm._map(x.coordinate_map_key)
cm.build_pyramid(x.coordinate_map_key, 6)


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


def line(name, rows, us, nbytes):
    print(f'{name:44s} {rows:10d} rows {us:9.1f} us  {nbytes / us / 1e6:6.2f} TB/s algorithmic ({nbytes / 1e6:8.1f} MB)', flush=True)


m1 = m0.parent                                   # stride 2: where the 3x3x3 MFMA layers run
p1 = cm._nbr27(m1.parent)
n0, n1 = m0.n, m1.n
# --- first layer: masks from the parent level, convolution of ones
pn0 = cm._nbr27(m1)
t = timed(lambda: ops.mask27_from_parent(m0.keys, m0.parent_of, pn0, m0.child_row))
line('k_nbr27_from_parent<mask> (finest)', n0, t, n0 * (8 + 4 + 4) + m1.n * (27 * 4 + 32))
masks0 = ops.mask27_from_parent(m0.keys, m0.parent_of, pn0, m0.child_row)
w = torch.randn(27, 1, 16, device='cuda') / 4
bias = torch.randn(16, device='cuda')
slope = torch.tensor([0.25], device='cuda')
t = timed(lambda: ops.conv_ones_k3(masks0, w, 16, bias=bias, act=ops.ACT_PRELU, slope=slope))
line('k_conv_ones_k3 (c_out 16)', n0, t, n0 * (4 + 64))
# --- neighbour table of the stride-2 level: table only, table + rows + masks, generated set
t = timed(lambda: ops.nbr27_from_parent(m1.keys, m1.parent_of, p1, m1.child_row))
line('k_nbr27_from_parent<table>', n1, t, n1 * (8 + 4 + 108) + m1.parent.n * (108 + 32))
t = timed(lambda: ops.nbr27_from_parent_ex(m1.keys, m1.parent_of, p1, m1.child_row))
line('k_nbr27_from_parent<table+rows+masks>', n1, t, n1 * (8 + 4 + 108 + 128 + 4) + m1.parent.n * (108 + 32))
g = cm._generated(m1)                            # the decoder's 8 M candidates
t = timed(lambda: ops.nbr27_from_parent(None, None, pn0, None, n=g.n), reps=5)
line('k_nbr27_from_parent<table> generated', g.n, t, g.n * 108 + m1.n * 108)
nbr, rows, masks = ops.nbr27_from_parent_ex(m1.keys, m1.parent_of, p1, m1.child_row)
t = timed(lambda: ops.transpose_table(nbr, 32))
line('k_transpose_table (rounds 2-5)', n1, t, n1 * (108 + 128))
t_keys_t = timed(lambda: ops.conv_row_order(nbr, 27, n1, 1, n1, 19))
t_keys_m = timed(lambda: ops.conv_row_order(None, 27, n1, 1, n1, 19, masks=masks))
print(f'conv_row_order: from the table {t_keys_t:.1f} us, from masks {t_keys_m:.1f} us', flush=True)
order = ops.conv_row_order(None, 27, n1, 1, n1, 19, masks=masks)
t = timed(lambda: rows.index_select(0, order.long()))
line('torch index_select rows (rounds 2-5)', n1, t, n1 * (4 + 128 + 128))
t = timed(lambda: ops.gather_table_rows(rows, order))
line('k_gather_table_rows', n1, t, n1 * (4 + 128 + 128))
# --- pyramid step and refinement
t = timed(lambda: ops.coarsen(m0.keys))
line('coarsen (flags + scan + k_coarsen_scatter)', n0, t, n0 * (8 + 4 + 4 + 4 + 4) + m1.n * 40)
mask = (torch.rand(8 * m1.n, device='cuda') < 0.45).to(torch.uint8)
t = timed(lambda: ops.refine(m1.keys, mask))
line('refine (scan + k_refine_scatter)', 8 * m1.n, t, 8 * m1.n * (1 + 4 + 4) + int(mask.sum()) * 12 + m1.n * 8)
# --- top-k threshold of the 8 M candidates of ONE cloud (per-cloud ranking)
mm = m1.n // B
logit = torch.randn(8 * mm, device='cuda')
t = timed(lambda: ops.topk_keep(logit, mm * 8 * 45 // 100))
line('topk_keep (select + mask), one cloud', 8 * mm, t, 8 * mm * (4 * 4 + 1))
# --- gather_sum and the classify head on the candidates
y = torch.randn(n1, 32, device='cuda')
t = timed(lambda: ops.gather_sum(y, nbr, 27, n1, 1, n1))
line('k_gather_sum (stride-2 level)', n1, t, n1 * (108 + 128 + 4))
xh = torch.randn(8 * m1.n, 16, device='cuda')
w1 = torch.randn(16, 8, device='cuda') / 4
b1 = torch.randn(8, device='cuda')
w2 = torch.randn(8, 1, device='cuda') / 3
b2 = torch.randn(1, device='cuda')
t = timed(lambda: ops.pointwise_head(xh, w1, b1, ops.ACT_PRELU, slope, 1, w2, b2), reps=5)
line('k_pointwise_head<16, 8> (the 8 M candidates)', xh.shape[0], t, xh.shape[0] * (64 + 4))
