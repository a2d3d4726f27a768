#!/usr/bin/env python3
"""per kernel: means of the SQ counters of one rocprofv3 --pmc pass (counter_collection.csv) as fractions of SQ_WAVE_CYCLES
    python tools/r06/sq_summary.py <counter_collection.csv>"""
import collections, csv, re, sys
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        m = re.search(r'(k_[a-z0-9_]+(<[^>]*>)?)', r['Kernel_Name'])
        if not m:
            continue
        k = m.group(1)
        acc[k][r['Counter_Name']] += float(r['Counter_Value'])
        cnt[(k, r['Counter_Name'])] += 1
names = sorted({c for v in acc.values() for c in v})
print('| kernel | launches | ' + ' | '.join(names) + ' |')
print('|---|---:|' + '---:|' * len(names))
for k, v in sorted(acc.items(), key=lambda kv: -kv[1].get('SQ_WAVE_CYCLES', 0)):
    wc = v.get('SQ_WAVE_CYCLES', 0) or 1
    n = max(cnt[(k, c)] for c in names)
    cells = [f"{v.get(c, 0) / 1e6:.1f} M" if c in ('SQ_WAVE_CYCLES', 'SQ_INSTS_VALU', 'SQ_INSTS_LDS', 'SQ_INSTS_VMEM_WR', 'SQ_INSTS_VMEM_RD') else f"{100 * v.get(c, 0) / wc:.1f} %" for c in names]
    print(f'| `{k}` | {n} | ' + ' | '.join(cells) + ' |')
