#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE cannot share a pass on gfx950).

    python profiles/pmc_summary.py <fetch counter_collection.csv> <write counter_collection.csv> [out.json]

Units and corrections (MI355X_MICROARCH.md, section HBM): both counters are in KiB; on gfx950 FETCH_SIZE reports half of
the bytes of a wide (16 B/lane) coalesced read, so it is doubled; WRITE_SIZE is exact for 16 B/lane stores.
Prints a markdown table and optionally writes {kernel: {launches, fetch_bytes, write_bytes, bytes_per_launch}}."""
import collections
import csv
import json
import re
import sys


def load(path):
    acc = collections.defaultdict(lambda: [0, 0.0])
    with open(path) as f:
        for r in csv.DictReader(f):
            m = re.search(r'fpcc::\(anonymous namespace\)::(k_[a-z0-9_]+(<[^>]*>)?)', r['Kernel_Name'])
            key = m.group(1) if m else 'non-fpcc kernels'
            acc[key][0] += 1
            acc[key][1] += float(r['Counter_Value'])
    return acc


def main():
    fetch, write = load(sys.argv[1]), load(sys.argv[2])
    out = {}
    print('| kernel | launches | FETCH_SIZE sum (MiB) | x2 corrected (MiB) | WRITE_SIZE sum (MiB) | HBM MiB / launch |')
    print('|---|---:|---:|---:|---:|---:|')
    for k in sorted(fetch, key=lambda k: -(2 * fetch[k][1] + write.get(k, [0, 0])[1])):
        n, f = fetch[k]
        w = write.get(k, [0, 0.0])[1]
        per = (2 * f + w) / n / 1024
        print(f'| `{k}` | {n} | {f / 1024:.1f} | {2 * f / 1024:.1f} | {w / 1024:.1f} | {per:.2f} |')
        out[k] = {'launches': n, 'fetch_bytes': 2 * f * 1024, 'write_bytes': w * 1024,
                  'bytes_per_launch': (2 * f + w) * 1024 / n}
    if len(sys.argv) > 3:
        with open(sys.argv[3], 'w') as fo:
            json.dump(out, fo, indent=1)


if __name__ == '__main__':
    main()
