#!/usr/bin/env python3
"""Dispatch-gap statistics of the bench's timed steps from a rocprofv3 --kernel-trace csv of `bench.py`.

    python profiles/step_gaps.py <p_kernel_trace.csv> [title] [from=0.35] [to=0.70]

Looks at the window [from, to] of the trace's time span (the middle of the timed region for the command lines of tools/r04/final.sh),
counts the frames in it by the launches of `k_conv_ones_k3` (the encoder's first layer: exactly one per compressed frame) and reports,
PER FRAME: the time some kernel or copy was running (union of the intervals), the idle time, the gaps by length class, and the
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


def main(path, title='', lo=0.35, hi=0.70):
    ev = []
    with open(path) as f:
        for r in csv.DictReader(f):
            ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name'])))
    ev.sort()
    t0, t1 = ev[0][0], max(e[1] for e in ev)
    a, b = t0 + lo * (t1 - t0), t0 + hi * (t1 - t0)
    win = [e for e in ev if e[0] >= a and e[1] <= b]
    frames = sum(1 for e in win if e[2] == 'k_conv_ones_k3')
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
    print(f'window: {span / 1e6:.1f} ms of the trace ({lo:.2f} .. {hi:.2f} of its span), {len(win)} dispatches, **{frames} frames** '
          f'(launches of `k_conv_ones_k3`)')
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
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else '', float(sys.argv[3]) if len(sys.argv) > 3 else 0.35,
         float(sys.argv[4]) if len(sys.argv) > 4 else 0.70)
