#!/usr/bin/env python3
"""Achieved HBM bandwidth of the path's bandwidth-bound helper kernels by launch size: all launches, the largest sixth, the largest one.
The per-kernel averages of hbm_bandwidth.py are dominated by the latency-sized launches on the small pyramid levels.

    python profiles/hbm_bandwidth_by_size.py <fetch dir> <write dir>     (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes with --kernel-trace)"""
import collections
import csv
import os
import sys

# (round 6: the table producer's template argument is a mode -- 0 table, 1 masks only, 2 table + row-major copy + masks; rounds 2-5: <false> / <true>)
KERNELS = ('k_nbr27_from_parent<false>', 'k_nbr27_from_parent<true>', 'k_nbr27_from_parent<2>', 'k_nbr27_from_parent<1>', 'k_nbr27_from_parent<0>',
           'k_gather_sum_generated', 'k_gather_sum', 'k_gather_table_rows', 'k_coarsen_scatter', 'k_refine_scatter',
           'k_conv_row_keys', 'k_conv_row_keys_masks', 'k_pointwise_head', 'k_conv_ones_k3', 'k_unique_scatter', 'k_logit_to_prob16', 'k_child_mask')


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
            flat = name.replace(' ', '')
            at = flat.find(key)
            # the key must be the whole kernel name (k_gather_sum is not k_gather_sum_generated) or name + template arguments as given
            if at < 0 or not (key.endswith('>') or flat[at + len(key): at + len(key) + 1] in ('(', '<', '')):
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
