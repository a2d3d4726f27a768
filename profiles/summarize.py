#!/usr/bin/env python3
"""Condenses a rocprofv3 `*_kernel_stats.csv` into a short markdown table (kernel names shortened)."""
import csv
import re
import sys


def short(name: str) -> str:
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    m = re.search(r'fpcc::(k_[a-z0-9_]+(<[^>]*>)?)', name)
    if m:
        return 'fpcc::' + m.group(1)
    if 'rocprim' in name:
        for key in ('radix_sort_onesweep', 'radix_sort_block_sort', 'merge_sort_block_merge', 'scan_impl',
                    'init_lookback_scan_state', 'onesweep_histograms', 'onesweep_scan_histograms'):
            if key in name:
                return 'rocprim::' + key
        return 'rocprim::other'
    if 'at::native' in name:
        m = re.search(r'at::native::([A-Za-z_0-9]+)', name)
        extra = 'MinNan' if 'MinNan' in name else ('sum' if 'sum_functor' in name else '')
        return 'torch::' + (m.group(1) if m else 'kernel') + (f'<{extra}>' if extra else '')
    return name[:60]


def main(path, steps):
    rows = {}
    with open(path) as f:
        for r in csv.DictReader(f):
            k = short(r['Name'])
            e = rows.setdefault(k, [0, 0.0])
            e[0] += int(r['Calls'])
            e[1] += float(r['TotalDurationNs'])
    total = sum(v[1] for v in rows.values())
    print(f'| kernel | calls | total ms | avg us | % of GPU time | ms / step ({steps} steps incl. warm-up) |')
    print('|---|---:|---:|---:|---:|---:|')
    for k, (calls, ns) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
        print(f'| `{k}` | {calls} | {ns / 1e6:.3f} | {ns / calls / 1e3:.1f} | {100 * ns / total:.2f} | {ns / 1e6 / steps:.3f} |')
    print(f'| **all kernels** | | {total / 1e6:.3f} | | 100 | {total / 1e6 / steps:.3f} |')


if __name__ == '__main__':
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 1)
