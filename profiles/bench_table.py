#!/usr/bin/env python3
"""The table of DESIGN section 6 from the bench lines of a collection: python profiles/bench_table.py profiles/r05 [name ...]"""
import json, os, sys


def load(path):
    return json.loads(open(path).read().strip().splitlines()[-1])


def main(d, names):
    cols = []
    for n in names:
        path = os.path.join(d, 'final_bench.json' if n == 'default' else f'final_bench_{n}.json')
        if os.path.exists(path):
            cols.append((n, load(path)))
    head = '| | ' + ' | '.join(f"{n} (B = {r['config'].get('batch_clouds')}, D = {r['config'].get('frames_in_flight')}{', named stages' if r['config'].get('named_stages') else ''})"
                               for n, r in cols) + ' |'
    print(head)
    print('|---|' + '---:|' * len(cols))
    rows = [('`value`, Mpoints/s', lambda r: f"{r['value']:.1f}"),
            ('ms per step', lambda r: f"{r['ms_per_step']:.1f}"),
            ('conv family, ms per step (events)', lambda r: f"{r['roofline'].get('kernel_ms_per_step', 0):.1f}"),
            ('`roofline.frac` of 157.3 TFLOP/s (events)', lambda r: f"{r['roofline']['frac']:.3f}"),
            ('... at the measured shader clock', lambda r: f"{r['roofline'].get('frac_at_shader_clock', 0):.3f} ({r['roofline'].get('shader_clock_mhz', 0) / 1e3:.2f} GHz)"),
            ('`value_one_frame`, Mpoints/s', lambda r: f"{r.get('value_one_frame', 0):.1f}"),
            ('one frame: encode + decode ms', lambda r: f"{r['config'].get('one_frame_ms', 0):.1f}")]
    for label, fn in rows:
        print(f'| {label} | ' + ' | '.join(fn(r) for _, r in cols) + ' |')


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2:] or ['default', 'b16_d2_stages', 'b16_d3', 'b16_d1', 'b8_d2', 'b4_d2', 'b1_d2', 'b1_d1'])
