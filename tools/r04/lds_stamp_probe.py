#!/usr/bin/env python3
"""Per-wave phase times of k_conv_lds (knob 3 = 32: stamps; 60: stamps on the bare structure without DMAs, barrier and fragment reads).
usage: lds_stamp_probe.py [level=1] [c_in=128] [c_out=128] [row_blocks=2]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from fastpcc_amd import engine as ME, hipops as ops
from fastpcc_amd.synthetic import SCALE, batched, body_cloud

level = int(sys.argv[1]) if len(sys.argv) > 1 else 1
c_in = int(sys.argv[2]) if len(sys.argv) > 2 else 128
c_out = int(sys.argv[3]) if len(sys.argv) > 3 else 128
rb = int(sys.argv[4]) if len(sys.argv) > 4 else 2
frame = torch.from_numpy(batched(body_cloud(1024, SCALE[1024], seed=2))).cuda()
cm = ME.CoordinateManager(D=3)
x = ME.SparseTensor(torch.ones((frame.shape[0], 1), device='cuda'), coordinates=frame, coordinate_manager=cm)
m = cm._map(x.coordinate_map_key)
for _ in range(level):
    m = cm._ensure_parent(m)
n = m.n
nbr = cm._nbr27(m)
order = cm._row_order(m)
torch.manual_seed(0)
f = torch.randn((n, c_in), device='cuda')
wt = torch.randn((27, c_in, c_out), device='cuda') / (13 * c_in) ** 0.5
run = lambda: ops.conv_f32(f, wt, c_out, n, row_order=order, pack=True, **cm._k3_table(m, os.environ.get('ROWS', '1') != '0', order))
ops.conv_set_tuning(ops.KNOB_LDS_ROWS, 1)
ops.conv_set_tuning(ops.KNOB_LDS_ROW_BLOCKS, rb)
S = 48
waves_per_wg = 2 * rb
n_waves = ((n + 32 * rb - 1) // (32 * rb)) * waves_per_wg
for dbg in (32, 60):
    ops.conv_set_tuning(ops.KNOB_WAVE_DBG, dbg)
    buf = torch.zeros(n_waves * S, dtype=torch.int64, device='cuda')
    ops.conv_debug_stamps(buf)
    for _ in range(60):
        run()
    buf.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(); e1.record()
    torch.cuda.synchronize()
    ops.conv_debug_stamps(None)
    st = buf.cpu().numpy().reshape(-1, S)
    st = st[st[:, 0] != 0]
    comp, stages = st[:, 43], st[:, 44]
    d = lambda a, b: (st[:, b] - st[:, a]).astype(np.float64)
    life = d(0, 42)
    rt = d(38, 39)
    clk = life[rt > 0] / rt[rt > 0] * 100.0
    print(f'  shader clock over a wave\'s life (memtime / memrealtime): median {np.median(clk):.0f} MHz  p10 {np.percentile(clk, 10):.0f}  p90 {np.percentile(clk, 90):.0f}')
    # occupancy over time from the constant 100 MHz counter (chip-wide): waves alive at 24 points of the launch
    ok = st[:, 39] > st[:, 38]
    b, e = st[ok, 38], st[ok, 39]
    t0, t1 = b.min(), e.max()
    pts = np.linspace(t0, t1, 25)[:-1] + (t1 - t0) / 48
    alive = [int(((b <= q) & (e > q)).sum()) for q in pts]
    print(f'  launch span by the 100 MHz counter: {(t1 - t0) / 100:.1f} us; waves alive at 24 points (1024 SIMDs x 3 = 3072): {alive}')
    first = np.sort(b)[[0, len(b) // 100, len(b) // 10]] - t0
    print(f'  wave starts: 1 % after {first[1] / 100:.1f} us, 10 % after {first[2] / 100:.1f} us; sum of wave lives {(e - b).sum() / 100 / 3072:.1f} us per wave slot')
    print(f'  wave life: sum {life.sum():.4g} cycles = {life.sum() / 1024 / 2400:.1f} us per SIMD slot-sum (1024 SIMDs, 2.4 GHz); median {np.median(life):.0f} p90 {np.percentile(life, 90):.0f}')
    for name, v in (('entry -> masks known', d(0, 1)), ('-> first DMAs issued', d(1, 2)), ('stage loop', d(2, 40)), ('fold', d(40, 41)), ('stores', d(41, 42))):
        print(f'  {name}: median {np.median(v):.0f}  p90 {np.percentile(v, 90):.0f}  share of wave life {v.sum() / life.sum() * 100:.1f} %')
    mf = comp.astype(np.float64) * 32 * 64
    print(f'  stages computed per wave: median {np.median(comp):.0f}; of the workgroup: median {np.median(stages):.0f}; executed MFMA cycles {mf.sum():.4g} '
          f'= {mf.sum() / 1024 / 2400:.1f} us per SIMD at 2.4 GHz')
    ok = comp > 0
    ratio = d(2, 40)[ok] / mf[ok]
    print(f'  stage loop / own MFMA cycles: median {np.median(ratio):.2f}  p10 {np.percentile(ratio, 10):.2f}  p90 {np.percentile(ratio, 90):.2f}   (3 waves share a SIMD: 3.0 = pipe always busy)')
    per_stage = d(2, 40)[ok] / stages[ok]
    print(f'  stage loop / stages of the workgroup: median {np.median(per_stage):.0f} cycles per stage')
ops.conv_set_tuning(ops.KNOB_WAVE_DBG, 0)
