#!/usr/bin/env python3
"""Achieved HBM bandwidth per kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) and the kernel trace of the
FETCH_SIZE pass:  GB/s = (2 * FETCH_SIZE + WRITE_SIZE) / kernel time  (units / gfx950 correction as in pmc_summary.py).

    python profiles/hbm_bandwidth.py <fetch dir> <write dir>      (directories holding p_counter_collection.csv, p_kernel_trace.csv)"""
import collections
import csv
import os
import re
import sys

PEAK = 8000.0


def key_of(name):
    m = re.search(r'fpcc::\(anonymous namespace\)::(k_[a-z0-9_]+(<[^>]*>)?)', name)
    if m:
        return m.group(1)
    return 'rocprim (sort / scan)' if 'rocprim' in name else None


def counters(d):
    acc = collections.defaultdict(float)
    with open(os.path.join(d, 'p_counter_collection.csv')) as f:
        for r in csv.DictReader(f):
            k = key_of(r['Kernel_Name'])
            if k:
                acc[k] += float(r['Counter_Value'])
    return acc


def main():
    fetch, write = counters(sys.argv[1]), counters(sys.argv[2])
    dur, n = collections.defaultdict(float), collections.Counter()
    with open(os.path.join(sys.argv[1], 'p_kernel_trace.csv')) as f:
        for r in csv.DictReader(f):
            k = key_of(r['Kernel_Name'])
            if k:
                dur[k] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
                n[k] += 1
    print('| kernel | launches | fetched MB (x2) | written MB | total ms | GB/s | % of 8 TB/s |')
    print('|---|---:|---:|---:|---:|---:|---:|')
    for k in sorted(dur, key=lambda k: -dur[k]):
        fb, wb = 2 * fetch.get(k, 0.0) * 1024, write.get(k, 0.0) * 1024
        gbs = (fb + wb) / (dur[k] * 1e-3) / 1e9 if dur[k] else 0.0
        print(f'| `{k}` | {n[k]} | {fb / 1e6:.1f} | {wb / 1e6:.1f} | {dur[k]:.3f} | {gbs:.0f} | {100 * gbs / PEAK:.1f} |')


if __name__ == '__main__':
    main()
