#!/usr/bin/env python3
"""Matrix-pipe utilisation per kernel inside the bench step from one rocprofv3 --pmc pass:
    busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs)         (both summed over the XCDs by rocprofv3)
    executed TFLOP per launch = SQ_INSTS_VALU_MFMA_MOPS_F32 / 8 MFMAs x 4096 flop
python profiles/mfma_busy.py <dir with p_counter_collection.csv>"""
import collections
import csv
import os
import re
import sys


def main(d):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.Counter()
    with open(os.path.join(d, 'p_counter_collection.csv')) as f:
        for r in csv.DictReader(f):
            m = re.search(r'fpcc::\(anonymous namespace\)::(k_[a-z0-9_]+(<[^>]*>)?)', r['Kernel_Name'])
            if not m:
                continue
            k = m.group(1)
            acc[k][r['Counter_Name']] += float(r['Counter_Value'])
            if r['Counter_Name'] == 'GRBM_GUI_ACTIVE':
                n[k] += 1
    print('| kernel | launches | GUI cycles per launch (one XCD) | matrix pipe busy | executed MFMA TFLOP per launch |')
    print('|---|---:|---:|---:|---:|')
    rows = [(v.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0), k, v) for k, v in acc.items() if v.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) > 0]
    tot_busy = tot_cyc = 0.0
    for busy, k, v in sorted(rows, reverse=True):
        gui = v['GRBM_GUI_ACTIVE'] / 8
        tot_busy += busy
        tot_cyc += gui * 1024
        print(f'| `{k}` | {n[k]} | {gui / n[k]:.0f} | {busy / (gui * 1024):.3f} | {v.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0) / 8 * 4096 / n[k] / 1e12:.4f} |')
    print(f'| **all MFMA kernels** | | | {tot_busy / tot_cyc:.3f} | |')


if __name__ == '__main__':
    main(sys.argv[1])
