#!/usr/bin/env python3
"""Rolls the per-launch table of `bench.py --dump-trace` (ONE traced step) up by map size and kernel shape.

    python profiles/conv_by_level.py <conv_launches.txt> > profiles/rNN/<tag>_conv_by_level.md"""
import collections
import sys

PEAK = 157.3
BANDS = [(200_000, '>= 200 K rows'), (50_000, '50-200 K rows'), (12_000, '12-50 K rows'), (3_000, '3-12 K rows'), (0, '< 3 K rows')]


def main(path):
    agg = collections.OrderedDict()
    for lo, name in BANDS:
        for shape in ('27 offsets', '8 offsets', '1 offset / fused chain'):
            agg[(name, shape)] = [0, 0.0, 0.0]
    with open(path) as f:
        next(f)
        for line in f:
            kind, c_in, c_out, n_out, n_off, groups, ms, gflop, *_ = line.split()
            if kind != 'mfma':
                continue
            band = next(name for lo, name in BANDS if int(n_out) >= lo)
            shape = '27 offsets' if int(n_off) == 27 else '8 offsets' if int(n_off) == 8 else '1 offset / fused chain'
            e = agg[(band, shape)]
            e[0] += 1
            e[1] += float(ms)
            e[2] += float(gflop)
    print(f'# MFMA convolution launches of the default bench run, by map size and kernel shape: ONE of the traced steps')
    print(f'# source: {path} (bench.py --dump-trace, the last traced step); peak {PEAK} TFLOP/s at the nominal 2.4 GHz; event-timed (includes launch gaps)')
    print('| rows of the map | kernel | launches | ms / step | algorithmic GFLOP | TFLOP/s | fraction of peak |')
    print('|---|---|---:|---:|---:|---:|---:|')
    tot = [0, 0.0, 0.0]
    for (band, shape), (n, ms, gf) in agg.items():
        if n:
            print(f'| {band} | {shape} | {n} | {ms:.3f} | {gf:.1f} | {gf / ms:.1f} | {gf / ms / PEAK:.3f} |')
            tot = [tot[0] + n, tot[1] + ms, tot[2] + gf]
    print(f'| **all** | | {tot[0]} | {tot[1]:.3f} | {tot[2]:.1f} | {tot[2] / tot[1]:.1f} | {tot[2] / tot[1] / PEAK:.3f} |')


if __name__ == '__main__':
    main(sys.argv[1])
