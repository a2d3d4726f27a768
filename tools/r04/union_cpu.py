#!/usr/bin/env python3
"""CPU study (numpy): executed / algorithmic MFMA work of a 3x3x3 layer when T consecutive rows of the pattern order share one
stage sequence (a workgroup tile walks the union of its rows' offsets; a wave of 32 rows skips the offsets it lacks)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from fastpcc_amd.synthetic import SCALE, body_cloud

def morton(p):
    def spread(v):
        v = v.astype(np.uint64) & 0x1fffff
        v = (v | (v << 32)) & 0x1f00000000ffff
        v = (v | (v << 16)) & 0x1f0000ff0000ff
        v = (v | (v << 8)) & 0x100f00f00f00f00f
        v = (v | (v << 4)) & 0x10c30c30c30c30c3
        v = (v | (v << 2)) & 0x1249249249249249
        return v
    return spread(p[:, 0]) | (spread(p[:, 1]) << 1) | (spread(p[:, 2]) << 2)

xyz = body_cloud(1024, SCALE[1024], seed=2)
for level in range(1, 4):
    xyz = np.unique(xyz >> 1, axis=0)
    key = morton(xyz)
    o = np.argsort(key); xyz = xyz[o]; key = key[o]
    n = len(xyz)
    masks = np.zeros(n, np.uint32)
    k = 0
    for dz in (-1, 0, 1):
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                q = xyz + np.array([dx, dy, dz])
                ok = (q >= 0).all(1)
                kk = morton(np.where(ok[:, None], q, 0))
                pos = np.searchsorted(key, kk)
                pos[pos >= n] = n - 1
                hit = ok & (key[pos] == kk)
                masks |= hit.astype(np.uint32) << k
                k += 1
    alg = sum(int(((masks >> k) & 1).sum()) for k in range(27))
    g = masks.copy()
    g ^= g >> 1; g ^= g >> 2; g ^= g >> 4; g ^= g >> 8; g ^= g >> 16
    for wlog in (17, 13):
        sk = ((np.arange(n, dtype=np.int64) >> wlog) << 32) | g.astype(np.int64)
        order = np.argsort(sk, kind='stable')
        m = masks[order]
        def unions(m, bs):
            pad = (-len(m)) % bs
            mm = np.concatenate((m, np.zeros(pad, np.uint32))).reshape(-1, bs)
            return np.bitwise_or.reduce(mm, axis=1)
        pc = lambda u: np.array([bin(int(x)).count('1') for x in u])
        u32 = unions(m, 32)
        line = f'level {level} rows {n} pairs/row {alg / n:.2f} window 2^{wlog}: wave-executed {pc(u32).sum() * 32 / alg:.3f}'
        for T in (64, 128, 256):
            uT = unions(m, T)
            line += f' | tile {T}: stage-time {pc(uT).sum() * T / alg:.3f}'
        print(line, flush=True)
        if wlog == 17:
            # second-level ordering: 32-row blocks re-sorted by (offsets lacking, Gray rank of the block's union mask); a tile = T/32 adjacent blocks
            nb = len(m) // 32
            ub = u32[:nb]
            gr = ub.copy(); gr ^= gr >> 1; gr ^= gr >> 2; gr ^= gr >> 4; gr ^= gr >> 8; gr ^= gr >> 16
            for name, keyb in (('weight,gray(union)', ((27 - pc(ub)).astype(np.int64) << 32) | gr.astype(np.int64)),
                               ('gray(union)', gr.astype(np.int64)),
                               ('weight only (current, 64-row groups ~)', ((27 - pc(ub)).astype(np.int64) << 32) | np.arange(nb))):
                bo = np.argsort(keyb, kind='stable')
                line = f'    blocks re-sorted by {name}: distinct unions {len(np.unique(ub))} of {nb} blocks'
                for T in (64, 128, 256):
                    q = T // 32
                    uu = ub[bo][: (nb // q) * q].reshape(-1, q)
                    uT = np.bitwise_or.reduce(uu, axis=1)
                    line += f' | tile {T}: stage-time {(pc(uT).sum() * T + (len(m) - (nb // q) * q * 32) * 14) / alg:.3f}'
                print(line, flush=True)
