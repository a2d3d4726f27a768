#!/usr/bin/env python3
"""The 27-offset 256 -> 256 int8 convolution (k_conv_i8_tiled) on every octree level of the LiDAR-like frame (cfg#3), alone:
time per launch, pairs, executed stages per tile, TOP/s, algorithmic bytes, % of the int8 MFMA peak and of 8 TB/s.
FPCC_I8_DBG selects the timing ablations of the kernel (results wrong).  usage: i8_probe.py [reps=30]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
from fastpcc_amd import hipops as ops
from fastpcc_amd.int_sparse_conv import _kernel_table, ROW_ORDER_WINDOW_LOG2
from fastpcc_amd.synthetic import lidar_cloud, batched
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
C = int(os.environ.get('CH', 256))
xyz = torch.from_numpy(batched(lidar_cloud(3))).cuda()
xyz = xyz - torch.nn.functional.pad(xyz.amin(0)[1:], (1, 0))
_, perm = ops.sort_keys(ops.morton3d_encode(xyz[:, 1:], (2, 1, 0)))
c = xyz[perm.long()].contiguous()
g = torch.Generator(device='cuda').manual_seed(0)
w = torch.randint(-127, 128, (27, C, C), dtype=torch.int8, device='cuda', generator=g)
mul = torch.full((C,), 1 << 10, dtype=torch.int64, device='cuda')
zp = torch.zeros(C, dtype=torch.int64, device='cuda')
bias = torch.zeros(C, dtype=torch.int32, device='cuda')
print('| rows | pairs/row | tiles x col tiles | stages per tile (128-row union x 2 chunks) | us | TOP/s | % of 5 POP/s | alg. MB | us at 8 TB/s | MFMA us per SIMD at peak |')
print('|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|')
for level in range(9):
    n = c.shape[0]
    if n < 2048:
        break
    _, table = _kernel_table(c, c, (3, 3, 3), (1, 1, 1), None)
    order = ops.conv_row_order((table - 1).contiguous(), 27, 1, 27, n, ROW_ORDER_WINDOW_LOG2)
    present = (table[:n] > 0)[order.long()]              # [n, 27] in tile order
    pairs = int(present.sum())
    pad = (-n) % 128
    tiles = torch.nn.functional.pad(present, (0, 0, 0, pad)).reshape(-1, 128, 27).any(1).sum(1)       # offsets per 128-row tile
    waves = torch.nn.functional.pad(present, (0, 0, 0, pad)).reshape(-1, 32, 27).any(1).sum()         # offsets per 32-row block
    a = torch.randint(-127, 128, (n, C), dtype=torch.int8, device='cuda', generator=g)
    kw = dict(nbr=table, n_offsets=27, nbr_ks=1, nbr_os=27, nbr_bias=1, bias=bias, requant_mul=mul, zero_point=zp, shift=18, out_bits=8,
              row_order=order)
    for _ in range(3):
        ops.conv_i8(a, w, C, C, n, **kw)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.conv_i8(a, w, C, C, n, **kw)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    ops_ = 2.0 * pairs * C * C
    alg = pairs * C + n * C + 27 * C * C + n * 27 * 4
    mfma_us = int(waves) * 2 * (C // 32) * 4 * 32 / 1024 / 2.4e3        # executed MFMAs (32 cycles each) spread over 1024 SIMDs at 2.4 GHz
    print(f'| {n} | {pairs / n:.2f} | {tiles.numel()} x {(C + 127) // 128} | {float(tiles.float().mean()) * 2 * ((C + 127) // 128 and 1):.1f} '
          f'| {us:.1f} | {ops_ / us / 1e6:.0f} | {ops_ / us / 1e6 / 5000 * 100:.1f} | {alg / 1e6:.1f} | {alg / 8e6:.1f} | {mfma_us:.1f} |')
    c = c.clone(); c[:, 1:] >>= 1
    c = torch.unique_consecutive(c, dim=0)
