#!/usr/bin/env python3
"""Host-side 255-ary rANS decode of cold CDF rows (the integer codec's decoder, fpcc_simple_dec_pop): ns per symbol with and without
the row warmers / the AVX-512 search.  Rows are made cold by streaming a 256-MB buffer between encode and decode.
usage: FPCC_HOST_WARMERS=k dec_bench.py [rows=200000]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from fastpcc_amd.rans_coder import RansDecoder, RansEncoder
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
rng = np.random.default_rng(0)
f = rng.integers(1, 400, (n, 255)).astype(np.int64)
f = (f * 65000 // f.sum(1, keepdims=True)) + 1
cdf = np.cumsum(f, 1)
cdf[:, -1] = 65535
rows = cdf.astype(np.uint16)
u = rng.integers(0, 65535, n)
sym = (rows <= u[:, None].astype(np.uint16)).sum(1).clip(0, 254).astype(np.uint16)
enc = RansEncoder(64 * 1024 * 1024)
enc.encode(rows, sym)
data = enc.flush()
evict = np.ones(256 * 1024 * 1024 // 8, dtype=np.int64)
best = None
for rep in range(5):
    evict += 1                                   # stream 256 MB: the rows leave every cache level
    dec = RansDecoder()
    dec.flush(data)
    out = np.empty(n, dtype=np.uint16)
    t0 = time.perf_counter()
    step = 16384
    for a in range(0, n, step):
        dec.decode(rows[a:a + step], out[a:a + step])
    dt = time.perf_counter() - t0
    assert (out == sym).all()
    best = dt if best is None else min(best, dt)
print(f'FPCC_HOST_WARMERS={os.environ.get("FPCC_HOST_WARMERS", "default")}: {best / n * 1e9:.1f} ns per symbol ({n} symbols, {len(data)} bytes)')
