#!/usr/bin/env python3
"""Dispatch-gap statistics of the bench's timed steps from a rocprofv3 --kernel-trace csv of `bench.py`.

    python profiles/step_gaps.py <p_kernel_trace.csv> [title] [from=0.40] [to=0.75]

Frames are counted by the launches of `k_conv_ones_k3` (the encoder's first layer: exactly one per compressed frame).  The window
runs from the start of frame number from x N to the start of frame number to x N of the N frames in the trace (the middle of the timed
region for the command lines of tools/r04/final.sh) and the tool reports, PER FRAME: the time some kernel or copy was running (union of the intervals), the idle time, the gaps by length class, and the
(kernel before -> kernel after) pairs that hold most of the idle time."""
import collections
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
    if 'copyBuffer' in name:
        return 'copy'
    if 'fillBuffer' in name:
        return 'fill'
    m = re.search(r'at::native::([A-Za-z_0-9]+)', name)
    return 'torch::' + m.group(1) if m else name[:40]


def main(path, title='', lo=0.40, hi=0.75):
    ev = []
    with open(path) as f:
        for r in csv.DictReader(f):
            ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name'])))
    ev.sort()
    marks = [e[0] for e in ev if e[2] == 'k_conv_ones_k3']
    if len(marks) < 4:
        raise SystemExit(f'only {len(marks)} frames in the trace')
    fa, fb = int(lo * len(marks)), max(int(hi * len(marks)), int(lo * len(marks)) + 1)
    a, b = marks[fa], marks[fb]
    win = [e for e in ev if e[0] >= a and e[0] < b]
    frames = fb - fa
    busy, cur_end, prev = 0, win[0][0], None
    gaps = []
    for s, e, nme in win:
        if s > cur_end:
            if prev is not None:
                gaps.append((s - cur_end, prev, nme))
            busy += e - s
            cur_end = e
        elif e > cur_end:
            busy += e - cur_end
            cur_end = e
        prev = nme
    span = cur_end - win[0][0]
    print(f'## {title}' if title else '## step gaps')
    print()
    print(f'window: {span / 1e6:.1f} ms, from the start of frame {fa} to the start of frame {fb} of the {len(marks)} in the trace: {len(win)} dispatches, '
          f'**{frames} frames**')
    print()
    print(f'per frame: **{span / 1e6 / frames:.2f} ms**, of which some kernel or copy runs {busy / 1e6 / frames:.2f} ms and the GPU is idle '
          f'**{(span - busy) / 1e6 / frames:.2f} ms** ({100 * (span - busy) / span:.1f} %); {len(win) / frames:.0f} dispatches')
    print()
    print('| gap length | gaps per frame | idle ms per frame |')
    print('|---|---:|---:|')
    for lo_us, hi_us in ((0, 5), (5, 20), (20, 100), (100, 500), (500, 10 ** 9)):
        g = [x[0] for x in gaps if lo_us * 1e3 <= x[0] < hi_us * 1e3]
        label = f'{lo_us} - {hi_us} us' if hi_us < 10 ** 9 else f'> {lo_us} us'
        print(f'| {label} | {len(g) / frames:.1f} | {sum(g) / 1e6 / frames:.3f} |')
    pairs = collections.defaultdict(lambda: [0, 0])
    for g, p, n in gaps:
        if g >= 20_000:
            pairs[(p, n)][0] += 1
            pairs[(p, n)][1] += g
    print()
    print('| gaps >= 20 us: kernel before -> kernel after | per frame | idle ms per frame | longest us |')
    print('|---|---:|---:|---:|')
    for (p, n), (cnt, tot) in sorted(pairs.items(), key=lambda kv: -kv[1][1])[:12]:
        longest = max(g for g, pp, nn in gaps if pp == p and nn == n)
        print(f'| `{p}` -> `{n}` | {cnt / frames:.1f} | {tot / 1e6 / frames:.3f} | {longest / 1e3:.0f} |')
    print()


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else '', float(sys.argv[3]) if len(sys.argv) > 3 else 0.40,
         float(sys.argv[4]) if len(sys.argv) > 4 else 0.75)
