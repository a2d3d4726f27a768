#!/usr/bin/env python3
"""Integer codec (cfg#3): host time of the phases of compress() / decompress() (no synchronisation added: what the host spends
issuing each phase) and the wall clock of each with a synchronise behind it."""
import os, sys, time, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from fastpcc_amd import replicas; replicas.bind_to_device_numa_node(0)       # as bench.py does
from fastpcc_amd.codecs.lossl_coord_int import Model, Config
from fastpcc_amd.codecs.lossl_coord_int import model as M
from fastpcc_amd.codecs.lossl_coord_int.init_random import randomize_
from fastpcc_amd.synthetic import lidar_cloud, batched
model = Model(Config(), 'cuda'); randomize_(model, 1); model = model.cuda().eval()
frame = torch.from_numpy(batched(lidar_cloud(3))).cuda()
SYNC = len(sys.argv) > 1 and sys.argv[1] == 'sync'
acc = {}


def timed(name, fn):
    def w(*a, **k):
        t0 = time.perf_counter()
        r = fn(*a, **k)
        if SYNC:
            torch.cuda.synchronize()
        acc.setdefault(name, []).append(time.perf_counter() - t0)
        return r
    return w


model.analyse = timed('analyse', model.analyse)
for blk in list(model.blocks_dec) + [model.block_dec_recurrent]:
    blk.compress = timed('level ' + type(blk).__name__, blk.compress)
    blk.decompress = timed('dlevel ' + type(blk).__name__, blk.decompress)
model.rans_decode_oct = timed('rans_decode_oct', model.rans_decode_oct)
M.ops.logits_to_ranges = timed('logits_to_ranges', M.ops.logits_to_ranges)
enc = model.rans_encoder
enc_ranges = enc.encode_ranges
tot = {'enc': [], 'dec': []}
for it in range(8):
    acc.clear()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    data = model.compress(frame); torch.cuda.synchronize(); t1 = time.perf_counter()
    rec = model.decompress(data); torch.cuda.synchronize(); t2 = time.perf_counter()
    if it >= 3:
        tot['enc'].append(t1 - t0); tot['dec'].append(t2 - t1)
        last = {k: list(v) for k, v in acc.items()}
print(f"enc {1e3*statistics.median(tot['enc']):.2f} ms  dec {1e3*statistics.median(tot['dec']):.2f} ms   ({'synchronised phases' if SYNC else 'host issue time'})")
for k, v in last.items():
    print(f'{k:40s} calls {len(v):3d}  total {1e3*sum(v):7.3f} ms   ' + ' '.join(f'{1e6*x:.0f}' for x in v))
