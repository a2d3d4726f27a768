#!/usr/bin/env python3
"""kernel durations of k_conv_i8_tiled by grid size from a rocprofv3 --kernel-trace csv.  usage: i8_trace_parse.py trace.csv [name substring]"""
import csv, sys, collections
rows = collections.defaultdict(list)
want = sys.argv[2] if len(sys.argv) > 2 else 'k_conv_i8_tiled'
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        if want in r['Kernel_Name']:
            g = int(r.get('Grid_Size_X', r.get('Grid_Size', 0))) // max(1, int(r.get('Workgroup_Size_X', r.get('Workgroup_Size', 1))))
            gy = int(r.get('Grid_Size_Y', 1))
            rows[(g, gy)].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k in sorted(rows, reverse=True):
    v = sorted(rows[k])
    print(f'grid {k[0]} x {k[1]}: {len(v)} launches, median {v[len(v) // 2]:.1f} us, min {v[0]:.1f} us')
