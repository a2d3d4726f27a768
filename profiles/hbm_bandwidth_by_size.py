#!/usr/bin/env python3
"""Achieved HBM bandwidth of the path's bandwidth-bound helper kernels by launch size: all launches, the largest sixth, the largest one.
The per-kernel averages of hbm_bandwidth.py are dominated by the latency-sized launches on the small pyramid levels.

    python profiles/hbm_bandwidth_by_size.py <fetch dir> <write dir>     (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes with --kernel-trace)"""
import collections
import csv
import os
import sys

KERNELS = ('k_nbr27_from_parent<false>', 'k_nbr27_from_parent<true>', 'k_gather_sum', 'k_coarsen_scatter', 'k_refine_scatter',
           'k_conv_row_keys', 'k_pointwise_head', 'k_conv_ones_k3', 'k_logit_to_prob16', 'k_child_mask')


def counters(d):
    by = collections.defaultdict(list)
    with open(os.path.join(d, 'p_counter_collection.csv')) as f:
        for r in csv.DictReader(f):
            by[r['Kernel_Name']].append((int(r['Dispatch_Id']), float(r['Counter_Value'])))
    return {k: [v for _, v in sorted(rows)] for k, rows in by.items()}


def durations(d):
    by = collections.defaultdict(list)
    with open(os.path.join(d, 'p_kernel_trace.csv')) as f:
        for r in csv.DictReader(f):
            by[r['Kernel_Name']].append((int(r['Dispatch_Id']), (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3))
    return {k: [v for _, v in sorted(rows)] for k, rows in by.items()}


def main(fetch_dir, write_dir):
    fetch, write, dur = counters(fetch_dir), counters(write_dir), durations(fetch_dir)
    print('| kernel | launches | all launches, TB/s | largest sixth, TB/s | largest launch | % of 8 TB/s (largest) |')
    print('|---|---:|---:|---:|---|---:|')
    for key in KERNELS:
        for name in fetch:
            if key + '(' not in name.replace(' ', '').replace('>(', '>(') and key not in name:
                continue
            rows = sorted(((2 * f + w) * 1024, us) for f, w, us in zip(fetch[name], write.get(name, []), dur.get(name, [])))[::-1]
            if not rows:
                continue
            top = rows[:max(1, len(rows) // 6)]
            bw = lambda rs: sum(b for b, _ in rs) / sum(t for _, t in rs) / 1e6
            print(f'| `{key}` | {len(rows)} | {bw(rows):.2f} | {bw(top):.2f} | {rows[0][0] / 1e6:.1f} MB in {rows[0][1]:.1f} us | {100 * bw(rows[:1]) / 8:.0f} |')
            break


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2])
