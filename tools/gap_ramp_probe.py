"""How much of the clock ramp does an idle gap cost?  Blocks of 24 back-to-back launches (64 -> 64 on the 272 K-row map, ~8 ms per
block) separated by GAP_MS of idle GPU; per block the mean launch time of its first and last four launches."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fastpcc_amd import engine as ME, hipops as ops
from fastpcc_amd.synthetic import SCALE, batched, body_cloud

frame = torch.from_numpy(batched(body_cloud(1024, SCALE[1024], seed=2))).cuda()
cm = ME.CoordinateManager(D=3)
x = ME.SparseTensor(torch.ones((frame.shape[0], 1), device='cuda'), coordinates=frame, coordinate_manager=cm)
m = cm._ensure_parent(cm._map(x.coordinate_map_key))
n = m.n
nbr, order = cm._nbr27(m), cm._row_order(m)
f = torch.randn((n, 64), device='cuda')
w = torch.randn((27, 64, 64), device='cuda') / (13 * 64) ** 0.5
out = torch.empty((n, 64), device='cuda')
fn = lambda: ops.conv_f32(f, w, 64, n, nbr=nbr, n_offsets=27, nbr_ks=n, nbr_os=1, row_order=order, pack=True, out=out)
fn(); fn()
torch.cuda.synchronize()
B, L = 12, 24
for gap_ms in (0.0, 0.5, 2.0, 5.0):
    time.sleep(0.1)
    res = []
    for b in range(B):
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(L + 1)]
        evs[0].record()
        for i in range(L):
            fn()
            evs[i + 1].record()
        torch.cuda.synchronize()
        ts = [evs[i].elapsed_time(evs[i + 1]) * 1e3 for i in range(L)]
        res.append((sum(ts[:4]) / 4, sum(ts[-4:]) / 4))
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < gap_ms * 1e-3:
            pass
    print(f'gap {gap_ms:3.1f} ms: first4/last4 us per block: ' + '  '.join(f'{a:.0f}/{b:.0f}' for a, b in res))
