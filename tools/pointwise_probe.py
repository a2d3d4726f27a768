#!/usr/bin/env python3
"""Per-point linear layer out = prelu(X @ W + b) on n rows in isolation: workgroup-tiled kernel (plain weights), wave kernel
and the persistent per-point kernel (packed weights).  usage: pointwise_probe.py [n] [c_in] [c_out] [reps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from fastpcc_amd import hipops as ops
n = int(sys.argv[1]) if len(sys.argv) > 1 else 272431
c_in = int(sys.argv[2]) if len(sys.argv) > 2 else 128
c_out = int(sys.argv[3]) if len(sys.argv) > 3 else 128
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 30
torch.manual_seed(0)
x = torch.randn((n, c_in), device='cuda')
w = torch.randn((c_in, c_out), device='cuda') / c_in ** 0.5
b = torch.randn(c_out, device='cuda')
slope = torch.tensor([0.2], device='cuda')


def timed(label, **kw):
    for _ in range(3):
        y = ops.conv_f32(x, w, c_out, n, bias=b, act=ops.ACT_PRELU, slope=slope, **kw)
    start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    start.record()
    for _ in range(reps):
        y = ops.conv_f32(x, w, c_out, n, bias=b, act=ops.ACT_PRELU, slope=slope, **kw)
    stop.record(); torch.cuda.synchronize()
    dt = start.elapsed_time(stop) / reps * 1e-3
    print(f'n {n} {c_in}->{c_out} [{label}]: {dt * 1e6:.1f} us  {2 * n * c_in * c_out / dt / 1e12:.1f} TFLOP/s  '
          f'{4 * n * (c_in + c_out) / dt / 1e12:.2f} TB/s')
    return y


base = timed('tiled')
ops.conv_set_tuning(ops.KNOB_POINTWISE_ROWS, 0)
wave = timed('wave', pack=True)
ops.conv_set_tuning(ops.KNOB_POINTWISE_ROWS, 1)
pw = timed('persistent', pack=True)
assert torch.equal(base, wave) and torch.equal(base, pw)
for dbg, what in ((1, 'no A traffic'), (4, 'no stores'), (5, 'neither'), (12, 'no loads at all, no stores')):
    ops.conv_set_tuning(ops.KNOB_WAVE_DBG, dbg)
    timed('persistent, ' + what, pack=True)
ops.conv_set_tuning(ops.KNOB_WAVE_DBG, 0)
