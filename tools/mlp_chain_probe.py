"""Times fpcc_mlp_chain_f32 against the layer-by-layer launches on the row counts of the cfg#2 pyramid.
    python tools/mlp_chain_probe.py [iters]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fastpcc_amd import hipops as ops

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
CHAINS = {'decoder block 1->64->128 ++128 ->128->128': (1, [64, 128, 128, 128], 2, 128),
          'decoder block 128->128->128': (128, [128, 128], -1, 0),
          'decoder block 64->64->64': (64, [64, 64], -1, 0)}


def timed(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


for name, (cx, widths, cat, cy) in CHAINS.items():
    for n in (272431, 70509, 17857, 4471, 1111, 279):
        rng = np.random.default_rng(n)
        x = torch.from_numpy(rng.normal(size=(n, cx)).astype(np.float32)).cuda()
        y = torch.from_numpy(rng.normal(size=(n, cy)).astype(np.float32)).cuda() if cy else None
        layers, c_in, flops = [], cx, 0
        for l, c in enumerate(widths):
            if l == cat:
                c_in += cy
            w = torch.from_numpy((rng.normal(size=(c_in, c)) / np.sqrt(c_in)).astype(np.float32)).cuda()
            layers.append((w, torch.zeros(c, device='cuda'), 1, torch.tensor([0.2], device='cuda'), 0.0))
            flops += 2 * n * c_in * c
            c_in = c

        def unfused():
            h = x
            for l, (w, b, act, s, clip) in enumerate(layers):
                h = ops.conv_f32(h, w, w.shape[1], n, x2=y if l == cat else None, bias=b, act=act, slope=s, clip=clip, pack=True)
            return h

        fused = lambda: ops.mlp_chain(x, layers, y=y, cat_layer=cat)
        assert torch.equal(fused(), unfused())
        tu = timed(unfused)
        ops.mlp_chain_set_form(0)
        tw = timed(fused)
        ops.mlp_chain_set_form(1)
        tf = timed(fused)
        print(f'{name:45s} rows {n:7d}: separate {tu:8.1f} us   wave form {tw:8.1f} us   workgroup form {tf:8.1f} us '
              f'({flops / tf / 1e6:6.1f} TFLOP/s)   x{tu / tf:.2f}')
