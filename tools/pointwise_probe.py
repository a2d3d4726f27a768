#!/usr/bin/env python3
"""Per-point linear layer out = prelu(X @ W + b) on n rows in isolation.  usage: pointwise_probe.py [n] [c_in] [c_out] [reps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from fastpcc_amd import hipops as ops
n = int(sys.argv[1]) if len(sys.argv) > 1 else 272431
c_in = int(sys.argv[2]) if len(sys.argv) > 2 else 128
c_out = int(sys.argv[3]) if len(sys.argv) > 3 else 128
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 20
torch.manual_seed(0)
x = torch.randn((n, c_in), device='cuda')
w = torch.randn((c_in, c_out), device='cuda') / c_in ** 0.5
b = torch.randn(c_out, device='cuda')
slope = torch.tensor([0.2], device='cuda')
for _ in range(3):
    y = ops.conv_f32(x, w, c_out, n, bias=b, act=ops.ACT_PRELU, slope=slope)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(reps):
    y = ops.conv_f32(x, w, c_out, n, bias=b, act=ops.ACT_PRELU, slope=slope)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
ref = torch.nn.functional.prelu(x.double() @ w.double() + b.double(), slope.double())
err = float((y.double() - ref).abs().max() / ref.abs().max())
print(f'n {n} {c_in}->{c_out}: {dt * 1e6:.1f} us  {2 * n * c_in * c_out / dt / 1e12:.1f} TFLOP/s  {4 * n * (c_in + c_out) / dt / 1e12:.2f} TB/s  rel err {err:.1e}')
