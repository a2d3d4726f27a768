#!/usr/bin/env python3
"""Stage-stamped runs of the grouped kernel (knob 3 = 16, fpcc_conv_debug_stamps) on the small pyramid levels of the headline cloud:
where a wave's time goes -- neighbour-table read, first operands, every (offset, chunk) stage, partial-sum exchange, stores.
usage: stamp_probe.py [levels=4,5,6] [c_in=128] [c_out=128]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from fastpcc_amd import engine as ME, hipops as ops
from fastpcc_amd.synthetic import SCALE, batched, body_cloud

levels = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else '4,5,6').split(',')]
c_in = int(sys.argv[2]) if len(sys.argv) > 2 else 128
c_out = int(sys.argv[3]) if len(sys.argv) > 3 else 128
frame = torch.from_numpy(batched(body_cloud(1024, SCALE[1024], seed=2))).cuda()
cm = ME.CoordinateManager(D=3)
x = ME.SparseTensor(torch.ones((frame.shape[0], 1), device='cuda'), coordinates=frame, coordinate_manager=cm)
m = cm._map(x.coordinate_map_key)
maps = {}
for lv in range(1, max(levels) + 1):
    m = cm._ensure_parent(m)
    maps[lv] = m
S = 48
for lv in levels:
    m = maps[lv]
    n = m.n
    nbr = cm._nbr27(m)
    order = cm._row_order(m)
    torch.manual_seed(0)
    f = torch.randn((n, c_in), device='cuda')
    w = torch.randn((27, c_in, c_out), device='cuda') / (13 * c_in) ** 0.5
    run = lambda: ops.conv_f32(f, w, c_out, n, nbr=nbr, n_offsets=27, nbr_ks=n, nbr_os=1, row_order=order, pack=True)
    def timed(reps=50):
        for _ in range(5):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(reps):
            run()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3
    ops.conv_set_tuning(ops.KNOB_WAVE_DBG, 0)
    ref = run().clone()
    t_plain = timed()
    print(f'## level {lv}: {n} rows, {c_in}->{c_out}, 27 offsets; plain build {t_plain:.1f} us per launch (back to back, incl. launch gap)')
    for dbg in (16, 17, 18, 19):
        ops.conv_set_tuning(ops.KNOB_WAVE_DBG, dbg)
        n_waves = ((n + 31) // 32) * (c_out // 32) * 4 * 2        # room for either column-group width
        buf = torch.zeros(n_waves * S, dtype=torch.int64, device='cuda')
        ops.conv_debug_stamps(buf)
        for _ in range(3):
            out = run()
        buf.zero_()
        out = run()
        torch.cuda.synchronize()
        t_st = timed(20)
        ops.conv_debug_stamps(None)
        if dbg == 16:
            assert torch.equal(out, ref), 'stamped build changed the result'
        st = buf.cpu().numpy().reshape(-1, S)
        st = st[st[:, 0] != 0]
        ns = st[:, 44]
        t0 = st[:, 0].min()
        span = (st[:, 42].max() - t0)
        mhz = 2400.0
        label = {16: 'real operands', 17: 'no gather traffic (every row = row 0)', 18: 'weights of one chunk (vector L1)', 19: 'neither'}[dbg]
        print(f'### dbg {dbg}: {label}; {len(st)} waves, stamped launch {t_st:.1f} us, first wave entry -> last store {span} cycles = {span / mhz:.1f} us at 2.4 GHz')
        d = lambda a, b: (st[:, b] - st[:, a])
        late = st[:, 0] - t0
        print(f'  wave entry after the first wave: median {np.median(late):.0f}  max {late.max()} cycles')
        for name, v in (('entry -> neighbour table in LDS', d(0, 1)), ('-> first operands requested', d(1, 2)),
                        ('loop (stage 0 top -> loop end)', st[:, 40] - st[:, 3]), ('loop end -> partial sums exchanged (barrier)', d(40, 41)),
                        ('-> outputs stored', d(41, 42)), ('whole wave', d(0, 42))):
            print(f'  {name}: median {np.median(v):.0f}  p90 {np.percentile(v, 90):.0f}  max {v.max()} cycles')
        has = ns > 0
        per = (st[has, 40] - st[has, 3]) / ns[has]
        print(f'  stages per wave: median {np.median(ns):.0f} max {ns.max()}; cycles per stage (loop / stages): median {np.median(per):.0f}  p10 {np.percentile(per, 10):.0f}  p90 {np.percentile(per, 90):.0f}   [16 MFMAs = 1024 cycles]')
        # by stage index: duration of stage i = top(i+1) - top(i)
        rows = []
        for i in range(0, 12):
            ok = ns > i + 1
            if ok.sum() == 0:
                break
            v = st[ok, 4 + i] - st[ok, 3 + i]
            rows.append(f'{np.median(v):.0f}')
        print('  median duration of stage 0, 1, 2, ...: ' + ' '.join(rows))
        wg = st[:, 45] >> 8
        xcc = st[:, 47] & 0xf
        cu = (st[:, 46] >> 8) & 0xf
        se = (st[:, 46] >> 13) & 0x7
        n_cu = len(set(zip(xcc.tolist(), se.tolist(), ((st[:, 46] >> 12) & 1).tolist(), cu.tolist())))
        print(f'  workgroups {len(set(wg.tolist()))} on {n_cu} distinct (xcc, se, sh, cu)')
    ops.conv_set_tuning(ops.KNOB_WAVE_DBG, 0)
