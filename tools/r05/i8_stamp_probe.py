#!/usr/bin/env python3
"""Stage-stamped runs of k_conv_i8_tiled (fpcc_conv_i8_debug_stamps) on octree levels of the LiDAR-like frame: where a workgroup's
life goes -- kernel-map slice, offset masks, the first operand round trip, every (offset, 128-channel chunk) stage, the epilogue.
Wave 0 of every workgroup stamps; medians over the workgroups of a launch, in s_memtime ticks (= shader cycles).
usage: i8_stamp_probe.py [levels=0,2,4,6]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from fastpcc_amd import hipops as ops
from fastpcc_amd.int_sparse_conv import _kernel_table, ROW_ORDER_WINDOW_LOG2
from fastpcc_amd.synthetic import lidar_cloud, batched
levels = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else '0,2,4,6').split(',')]
C, S = 256, 48
xyz = torch.from_numpy(batched(lidar_cloud(3))).cuda()
xyz = xyz - torch.nn.functional.pad(xyz.amin(0)[1:], (1, 0))
_, perm = ops.sort_keys(ops.morton3d_encode(xyz[:, 1:], (2, 1, 0)))
c = xyz[perm.long()].contiguous()
g = torch.Generator(device='cuda').manual_seed(0)
w = torch.randint(-127, 128, (27, C, C), dtype=torch.int8, device='cuda', generator=g)
mul = torch.full((C,), 1 << 10, dtype=torch.int64, device='cuda')
zp = torch.zeros(C, dtype=torch.int64, device='cuda')
bias = torch.zeros(C, dtype=torch.int32, device='cuda')
for level in range(max(levels) + 1):
    n = c.shape[0]
    if level in levels:
        _, table = _kernel_table(c, c, (3, 3, 3), (1, 1, 1), None)
        order = ops.conv_row_order((table - 1).contiguous(), 27, 1, 27, n, ROW_ORDER_WINDOW_LOG2)
        a = torch.randint(-127, 128, (n, C), dtype=torch.int8, device='cuda', generator=g)
        kw = dict(nbr=table, n_offsets=27, nbr_ks=1, nbr_os=27, nbr_bias=1, bias=bias, requant_mul=mul, zero_point=zp, shift=18, out_bits=8,
                  row_order=order)
        ref = ops.conv_i8(a, w, C, C, n, **kw).clone()
        n_wg = ((n + 127) // 128) * 2
        buf = torch.zeros(n_wg * S, dtype=torch.int64, device='cuda')
        ops.conv_i8_debug_stamps(buf)
        for _ in range(3):
            out = ops.conv_i8(a, w, C, C, n, **kw)
        buf.zero_()
        out = ops.conv_i8(a, w, C, C, n, **kw)
        torch.cuda.synchronize()
        ops.conv_i8_debug_stamps(None)
        assert torch.equal(out, ref), 'the stamped build changed the result'
        t = buf.cpu().numpy().reshape(n_wg, S).astype(np.int64)
        t = t[t[:, 0] > 0]
        stages = t[:, 44]
        med = lambda v: float(np.median(v))
        span = int(t[:, 39].max() - t[:, 0].min())
        life = t[:, 39] - t[:, 0]
        print(f'## level {level}: {n} rows, {len(t)} workgroups stamped, stages per workgroup median {med(stages):.0f} (max {stages.max()}); '
              f'kernel span {span} ticks; workgroup life median {med(life):.0f} ticks = {med(life) / span * 100:.0f} % of the span')
        rows = [('kernel-map slice -> LDS (row order, 7 x 16 B per row, barrier)', t[:, 1] - t[:, 0]),
                ('offset masks (27 LDS reads + ballots, barrier)', t[:, 2] - t[:, 1]),
                ('first operands: A gather + W tile -> LDS, barrier', t[:, 3] - t[:, 2])]
        st_all = []
        for i in range(len(t)):
            ns = int(min(stages[i], 32))
            if ns >= 1:
                tops = t[i, 4:4 + ns]
                ends = np.append(tops[1:], t[i, 38]) if ns < 33 else None
                st_all.extend((ends - tops).tolist())
        rows.append((f'one stage (fetch next, <= 16 MFMAs of 32 cycles, stash W, barrier); {len(st_all)} stages', np.array(st_all)))
        rows.append(('all stages of a workgroup', t[:, 38] - t[:, 3]))
        rows.append(('epilogue (64 elements per lane, byte stores) incl. the wait for the stores', t[:, 39] - t[:, 38]))
        print('| phase | median ticks | 10 % | 90 % | share of the workgroup life |')
        print('|---|---:|---:|---:|---:|')
        for name, v in rows:
            v = v[v >= 0] if len(v) else v
            if len(v) == 0:
                continue
            share = '' if name.startswith('one stage') else f'{np.sum(v) / np.sum(life) * 100:.0f} %'
            print(f'| {name} | {med(v):.0f} | {np.percentile(v, 10):.0f} | {np.percentile(v, 90):.0f} | {share} |')
        # how many workgroups are resident at a time: sum of lives / span
        print(f'resident workgroups on average: {np.sum(life) / span:.0f} of {len(t)} (2 per CU x 256 CUs = 512 slots)')
        print()
    c = c.clone(); c[:, 1:] >>= 1
    c = torch.unique_consecutive(c, dim=0)
