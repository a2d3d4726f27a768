#!/usr/bin/env python3
"""Dispatch-gap timeline of the timed steps from a rocprofv3 --kernel-trace csv of `bench.py`: per step the span from the first to the
last kernel, the time some kernel was running (union of intervals: with frames in flight two kernels never overlap on one stream, but
copies and probe kernels may), the idle time, and the longest gaps with the kernels either side of them.

    python profiles/step_gaps.py <p_kernel_trace.csv> [n_last_steps=3] [title]

A step boundary is the first launch of the encoder's first kernel (k_keys_from_coords on the input cloud)."""
import csv
import re
import sys


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    m = re.search(r'fpcc::(k_[a-z0-9_]+)', name)
    if m:
        return m.group(1)
    if 'rocprim' in name:
        return 'rocprim'
    m = re.search(r'at::native::([A-Za-z_0-9]+)', name)
    return 'torch::' + m.group(1) if m else name[:40]


def main(path, n_last=3, title=''):
    ev = []
    with open(path) as f:
        for r in csv.DictReader(f):
            ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name'])))
    ev.sort()
    # frames: every k_morton / first-kernel occurrence that follows a k_nn_dist2-free stretch; robust choice: the encoder's voxel keys of
    # the FULL cloud = the largest launches of k_keys_from_coords; simpler and exact for this bench: split at gaps > 1.5 ms after a decode
    starts = [0]
    for i in range(1, len(ev)):
        if ev[i][2] == 'k_keys_from_coords' and ev[i - 1][2] != 'k_keys_from_coords' and ev[i][0] - ev[starts[-1]][0] > 5_000_000:
            starts.append(i)
    starts.append(len(ev))
    frames = [(starts[i], starts[i + 1]) for i in range(len(starts) - 1)]
    print(f'# {title}' if title else '# step gaps')
    print()
    print(f'{len(ev)} kernel dispatches, {len(frames)} segments (a segment starts at the first `k_keys_from_coords` at least 5 ms after the previous segment\'s start)')
    print()
    print('| segment | dispatches | span ms | busy ms | idle ms | gaps > 20 us | idle in them ms | longest gaps (us: kernel before -> kernel after) |')
    print('|---:|---:|---:|---:|---:|---:|---:|---|')
    for fi, (a, b) in enumerate(frames[-n_last - 1:-1] if len(frames) > n_last + 1 else frames):
        seg = ev[a:b]
        span = (max(e[1] for e in seg) - seg[0][0]) / 1e6
        busy, cur_end, gaps = 0, seg[0][0], []
        for s, e, nme in seg:
            if s > cur_end:
                gaps.append((s - cur_end, prev, nme))
                busy += e - s
                cur_end = e
            else:
                if e > cur_end:
                    busy += e - cur_end
                    cur_end = e
            prev = nme
        big = [g for g in gaps if g[0] > 20_000]
        top = sorted(big, reverse=True)[:6]
        print(f'| {fi} | {len(seg)} | {span:.2f} | {busy / 1e6:.2f} | {span - busy / 1e6:.2f} | {len(big)} | {sum(g[0] for g in big) / 1e6:.2f} | '
              + '; '.join(f'{g[0] / 1e3:.0f}: {g[1]} -> {g[2]}' for g in top) + ' |')


if __name__ == '__main__':
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 3, sys.argv[3] if len(sys.argv) > 3 else '')
