"""Verdict item: the same convolution layer inside the traced bench step and alone.  One process, one box: code the cfg#2 frame with
per-launch events (what bench.py's roofline is made of), then re-launch the ten heaviest 3x3x3 layers in isolation on the SAME
coordinate maps (same neighbour tables, same row order, random features of the same shape), 20 back-to-back repetitions each,
and once more with 256 MB of unrelated traffic between repetitions (cold L2 / Infinity Cache).
    python tools/probe_vs_trace.py > profiles/r03/probe_vs_trace.md"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fastpcc_amd import engine as ME, hipops as ops
from fastpcc_amd.codecs.lossy_coord_v2 import Model
from fastpcc_amd.codecs.lossy_coord_v2.model_config import baseline_r1
from fastpcc_amd.synthetic import SCALE, batched, body_cloud, enliven

torch.manual_seed(0)
model = Model(baseline_r1())
enliven(model, 0)
model = model.cuda().eval()
frame = torch.from_numpy(batched(body_cloud(1024, SCALE[1024], seed=2))).cuda()


def step(keep_cm=False):
    data = model.compress(frame)
    torch.cuda.synchronize()
    if not keep_cm:
        ME.clear_global_coordinate_manager()
    rec = model.decompress(data)
    torch.cuda.synchronize()
    ME.clear_global_coordinate_manager()
    return data


for _ in range(4):
    step()
ops.reserve_trace_events(1500)
traces = []
for _ in range(3):
    ops.CONV_TRACE = []
    step()
    traces.append(ops.CONV_TRACE)
    ops.CONV_TRACE = None
# per (c_in, c_out, n_out): launches and their event times, median over the three traced steps
layers = {}
for ti, trace in enumerate(traces):
    for ev0, ev1, info in trace:
        if info['n_offsets'] != 27 or not info['mfma']:
            continue
        key = (info['c_in'], info['c_out'], info['n_out'])
        layers.setdefault(key, [[], [], []])[ti].append(ev0.elapsed_time(ev1) * 1e3)
rows = []
for key, per_step in layers.items():
    count = len(per_step[0])
    per_launch = sorted(sum(per_step, []))
    rows.append((sum(per_launch) / 3, key, count, per_launch[len(per_launch) // 2]))
rows.sort(reverse=True)

# the coordinate maps of the frame, rebuilt once (same generator: same tables)
cm = ME.CoordinateManager(D=3)
x = ME.SparseTensor(torch.ones((frame.shape[0], 1), device='cuda'), coordinates=frame - torch.nn.functional.pad(frame.amin(0)[1:], (1, 0)),
                    coordinate_manager=cm)
maps = {}
m = cm._map(x.coordinate_map_key)
while m is not None and m.n > 8:
    maps[m.n] = m
    try:
        m = cm._ensure_parent(m)
    except Exception:
        break
junk = torch.empty(64 * 1024 * 1024, dtype=torch.float32, device='cuda')


def timed(fn, reps, flush):
    for _ in range(3):
        fn()
    ts = []
    for _ in range(reps):
        if flush:
            junk.add_(1.0)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


def queued(fn, reps):
    """reps launches queued without a gap (the GPU never idles between them: sustained clocks, as inside the step)"""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps


print('# The ten heaviest 3x3x3 layers of the cfg#2 step: inside the traced step and alone (same box, same process)')
print('| layer | rows | launches / step | in the step, median us | alone, one launch at a time (GPU idle in between), us | alone, caches flushed, us '
      '| alone, 40 launches queued without a gap, us | step / one at a time | step / queued |')
print('|---|---:|---:|---:|---:|---:|---:|---:|---:|')
for total, (c_in, c_out, n), count, med in rows[:10]:
    m = maps.get(n)
    if m is None:
        continue
    nbr, order = cm._nbr27(m), cm._row_order(m)
    c1 = c_in if c_in <= 128 else 128
    x1 = torch.randn((n, c1), device='cuda')
    x2 = torch.randn((n, c_in - c1), device='cuda') if c_in > c1 else None
    w = torch.randn((27, c_in, c_out), device='cuda') / (13 * c_in) ** 0.5
    fn = lambda: ops.conv_f32(x1, w, c_out, n, x2=x2, nbr=nbr, n_offsets=27, nbr_ks=n, nbr_os=1, row_order=order, pack=True)
    alone, cold, sustained = timed(fn, 20, False), timed(fn, 20, True), queued(fn, 40)
    print(f'| {c_in} -> {c_out} | {n} | {count} | {med:.1f} | {alone:.1f} | {cold:.1f} | {sustained:.1f} | {med / alone:.3f} | {med / sustained:.3f} |')
