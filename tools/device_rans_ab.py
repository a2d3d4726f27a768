#!/usr/bin/env python3
"""A/B of the entropy-decoding stage: host coder pool (default) against the device-side rANS decoders, inside
`decompress` of lossy_coord_v2 (cfg#2, 1 M voxels) and lossl_coord_int (cfg#3, LiDAR sweep).  Prints a markdown record
(profiles/r02/device_rans.md)."""
import os, statistics, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from fastpcc_amd import engine as ME, hipops as ops
from fastpcc_amd.rans_coder import BinaryRansCoder
from fastpcc_amd.synthetic import SCALE, batched, body_cloud, enliven, lidar_cloud


def med(fn, reps=7, warm=2):
    ts = []
    for it in range(warm + reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = fn(); torch.cuda.synchronize()
        if it >= warm:
            ts.append(time.perf_counter() - t0)
        ME.clear_global_coordinate_manager()
    return statistics.median(ts) * 1e3, out


print('# Entropy decoding on the device vs on the host (MI355X, round 2)\n')
print('`tools/device_rans_ab.py`; median of 7 after 2 warm-ups, wall clock of `decompress` closed by a device synchronise.\n')

# raw decoders ---------------------------------------------------------------------------------------------------
print('## The serial chain alone\n')
print('| decoder | symbols | host (one core) | device (one wave) |')
print('|---|---:|---:|---:|')
rng = np.random.default_rng(0)
for n in (70_000, 564_000):
    p = np.clip(np.round(rng.beta(.3, .3, n) * 65536), 1, 65535).astype(np.uint16)
    bits = rng.random(n) < p / 65536
    coder = BinaryRansCoder(1)
    stream = coder.encode(bits[None], p[None].astype(np.uint32))[0]
    out = np.zeros((1, n), dtype=bool)
    t0 = time.perf_counter()
    for _ in range(5):
        coder.decode([stream], p[None].astype(np.uint32), out)
    host_ms = (time.perf_counter() - t0) / 5 * 1e3
    ds, dp = ops.stream_to_device(stream, 'cuda'), torch.from_numpy(p.view(np.int16)).cuda()
    dev_ms, _ = med(lambda: ops.rans_binary_decode_dev(ds, len(stream), dp))
    print(f'| binary (v2 occupancy level) | {n} | {host_ms:.2f} ms ({host_ms * 1e6 / n:.1f} ns/sym) | {dev_ms:.2f} ms ({dev_ms * 1e6 / n:.1f} ns/sym) |')
from fastpcc_amd.rans_coder import RansDecoder, RansEncoder
n = 100_000
f = rng.integers(1, 400, (n, 255)).astype(np.int64)
c = np.cumsum(f * (65000 // f.sum(1, keepdims=True)), 1); c[:, -1] = 65535
rows = c.astype(np.uint16); sym = rng.integers(0, 255, n).astype(np.uint16)
enc = RansEncoder(1 << 24); enc.encode(rows, sym); stream = enc.flush()
dec = RansDecoder(); got = np.zeros(n, np.uint16)
t0 = time.perf_counter()
for _ in range(3):
    dec.flush(stream); dec.decode(rows, got)
host_ms = (time.perf_counter() - t0) / 3 * 1e3
ds, dr = ops.stream_to_device(stream, 'cuda'), torch.from_numpy(rows.view(np.int16)).cuda()
x = int.from_bytes(stream[:4], 'little')
def run():
    st = torch.tensor([x - (1 << 32) if x >= 1 << 31 else x, 4, 0, 0], dtype=torch.int32).cuda()
    return ops.simple_dec_pop_dev(st, ds, len(stream), dr)
dev_ms, (s2, _) = med(run)
assert (s2.cpu().numpy().view(np.uint16) == sym).all()
print(f'| 255-ary rows (int codec level), rows already in host memory / in HBM | {n} | {host_ms:.2f} ms ({host_ms * 1e6 / n:.1f} ns/sym) | {dev_ms:.2f} ms ({dev_ms * 1e6 / n:.1f} ns/sym) |')

# inside the codecs -------------------------------------------------------------------------------------------------
print('\n## Inside `decompress`\n')
print('| codec / frame | host coder pool | device decoders |')
print('|---|---:|---:|')
from fastpcc_amd.codecs.lossy_coord_v2 import Model
from fastpcc_amd.codecs.lossy_coord_v2.model_config import baseline_r1
torch.manual_seed(0)
model = Model(baseline_r1()); enliven(model, 0); model = model.cuda().eval()
xyz = body_cloud(1024, SCALE[1024], seed=2)
frame = torch.from_numpy(batched(xyz)).cuda()
data = model.compress(frame); ME.clear_global_coordinate_manager()
h_ms, rec_h = med(lambda: model.decompress(data))
model.em_lossless_based.device_decoder = True
d_ms, rec_d = med(lambda: model.decompress(data))
model.em_lossless_based.device_decoder = False
assert torch.equal(rec_h, rec_d)
print(f'| lossy_coord_v2/baseline_r1, {len(xyz)} voxels (cfg#2), 6 occupancy levels | {h_ms:.2f} ms | {d_ms:.2f} ms |')
del model, frame
from fastpcc_amd.codecs.lossl_coord_int import Config, Model as IntModel
from fastpcc_amd.codecs.lossl_coord_int.init_random import randomize_
m = IntModel(Config(), 'cuda'); randomize_(m, 1); m = m.cuda().eval()
xyz = lidar_cloud(3)
frame = torch.from_numpy(batched(xyz)).cuda()
data = m.compress(frame)
h_ms, rec_h = med(lambda: m.decompress(data), reps=5)
m.device_decoder = True
d_ms, rec_d = med(lambda: m.decompress(data), reps=5)
m.device_decoder = False
assert torch.equal(rec_h, rec_d)
print(f'| lossl_coord_int, {len(xyz)} voxels (cfg#3), 13 levels, 510 B of CDF row per symbol | {h_ms:.2f} ms | {d_ms:.2f} ms |')
