#!/usr/bin/env python3
"""Host binary rANS coder, ns per symbol on occupancy-like probabilities (CPU only; used to judge coder changes).  Round 1, build
container's CPU: a tabulated-reciprocal quotient and a branch-free refill both measured slower than the divide / the byte loop.
Round 2, the GPU box's EPYC 9575F: quotient by a 128-bit multiply with a tabulated ceil(2^47 / freq) 2.90 -> 2.30 ns per encoded
symbol (same bytes); decode 2.44 -> 1.52 ns by forming both successor states side by side and picking one with a conditional
move (the chain through the state becomes shift -> multiply -> add -> cmov)."""
import sys, time, numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from fastpcc_amd._native import host, host_check
rng = np.random.default_rng(0)
n = 600000
# occupancy-like: most probabilities confident, some unsure
conf = rng.random(n) < 0.7
p = np.where(conf, rng.beta(0.3, 6, n), rng.random(n))
p1 = np.clip(np.rint(p * 65536), 1, 65535).astype(np.uint16)
bits = (rng.random(n) < p1 / 65536).astype(np.uint8)
cap = n + 64
out = np.empty(cap, np.uint8)
L = host()
ln = host_check(L.fpcc_rans_binary_encode(bits.ctypes.data, p1.ctypes.data, n, out.ctypes.data, cap))
stream = out[cap - ln:].copy()
dec = np.empty(n, np.uint8)
for name in ('decode', 'encode'):
    best = 1e9
    for _ in range(15):
        t = time.perf_counter()
        if name == 'decode':
            host_check(L.fpcc_rans_binary_decode(stream.ctypes.data, stream.size, p1.ctypes.data, n, dec.ctypes.data))
        else:
            host_check(L.fpcc_rans_binary_encode(bits.ctypes.data, p1.ctypes.data, n, out.ctypes.data, cap))
        best = min(best, time.perf_counter() - t)
    print(name, f'{best * 1e9 / n:.2f} ns/symbol', ln, 'bytes')
assert (dec == bits).all()
import hashlib; print(hashlib.sha256(stream.tobytes()).hexdigest()[:12])
