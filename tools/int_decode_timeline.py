#!/usr/bin/env python3
"""Where the decode wall clock of the integer codec goes: per level, waiting for the GPU (network + CDF kernel + D2H of the rows)
and the host rANS decode."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import torch
from fastpcc_amd import hipops as ops
from fastpcc_amd.codecs.lossl_coord_int import Model, Config
from fastpcc_amd.codecs.lossl_coord_int import model as M
from fastpcc_amd.codecs.lossl_coord_int.init_random import randomize_
from fastpcc_amd.synthetic import lidar_cloud, batched
xyz = lidar_cloud(3)
model = Model(Config(), 'cuda'); randomize_(model, 1); model = model.cuda().eval()
frame = torch.from_numpy(batched(xyz)).cuda()
data = model.compress(frame)
log = []
def timed(logits):
    t0 = time.perf_counter()
    rows_d = ops.logits_to_cdf16(logits.contiguous(), M.PRE_SHIFT)
    rows_h = torch.empty(rows_d.shape, dtype=rows_d.dtype, pin_memory=True)
    rows_h.copy_(rows_d, non_blocking=True)
    torch.cuda.current_stream().synchronize()
    t1 = time.perf_counter()
    out_h = torch.empty(rows_h.shape[0], dtype=torch.int16, pin_memory=True)
    model.rans_decoder.decode(rows_h.numpy().view(np.uint16), out_h.numpy().view(np.uint16))
    t2 = time.perf_counter()
    log.append((rows_h.shape[0], t0, t1, t2))
    return out_h.to(logits.device, non_blocking=True)
model.rans_decode_oct = timed
for it in range(4):
    log.clear()
    torch.cuda.synchronize(); T0 = time.perf_counter()
    model.decompress(data); torch.cuda.synchronize(); T1 = time.perf_counter()
wait = sum(t1 - t0 for _, t0, t1, _ in log); dec = sum(t2 - t1 for _, _, t1, t2 in log)
print(f'decode {1e3 * (T1 - T0):.2f} ms: waiting for GPU + D2H {1e3 * wait:.2f} ms, host rANS {1e3 * dec:.2f} ms, '
      f'host enqueue between levels {1e3 * (T1 - T0 - wait - dec):.2f} ms, {sum(n for n, *_ in log)} symbols')
prev = T0
for n, t0, t1, t2 in log:
    print(f'  {n:7d} symbols: enqueue {1e3 * (t0 - prev):6.2f}  wait {1e3 * (t1 - t0):6.2f}  rANS {1e3 * (t2 - t1):6.2f} ms ({1e9 * (t2 - t1) / n:5.1f} ns/symbol)')
    prev = t2
