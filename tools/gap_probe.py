"""Why is a 3x3x3 layer slower inside the step than alone?  Capture the EXACT arguments (tensors included) of the step's heaviest
conv_f32 calls, then replay each one right after the step: (a) as captured, (b) with random features in the captured buffers' place,
(c) with the captured features but a freshly built neighbour table / row order of the same map.  One process, one box."""
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


def step():
    data = model.compress(frame)
    torch.cuda.synchronize()
    ME.clear_global_coordinate_manager()
    model.decompress(data)
    torch.cuda.synchronize()
    ME.clear_global_coordinate_manager()


for _ in range(4):
    step()
ops.reserve_trace_events(1500)
captured = []
real = ops.conv_f32
clocks = torch.zeros((4096, 2), dtype=torch.int64, device='cuda')
n_clk = [0]


def probe_clock():
    ops.clock_probe(clocks[n_clk[0]], 10)
    n_clk[0] += 1
    return n_clk[0] - 1



def spy(x1, w, c_out, n_out, **kw):
    big = kw.get('n_offsets', 1) == 27 and n_out >= 60000 and ops.CONV_TRACE is not None
    ck = probe_clock() if big and CLOCKS else -1
    out = real(x1, w, c_out, n_out, **kw)
    if big:
        captured.append((len(ops.CONV_TRACE) - 1, x1, w, c_out, n_out, dict(kw), out, ck))
    return out


CLOCKS = bool(int(os.environ.get('CLOCKS', '1')))


ops.conv_f32 = spy
for mod in list(sys.modules.values()):
    if mod is not None and getattr(mod, 'conv_f32', None) is real and mod is not ops:
        mod.conv_f32 = spy
ops.CONV_TRACE = []
step()
trace = ops.CONV_TRACE
ops.CONV_TRACE = None
ops.conv_f32 = real


def queued(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    last_clock[0] = probe_clock()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps


last_clock = [0]


def mhz(i):
    c, t = clocks[i].tolist()
    return c / max(t, 1) * 100


print(f'captured {len(captured)} launches')
print('| # in step | layer | rows | ld1 | x2 | act | in the step us | shader clock before it, MHz | replay as captured | shader clock after the replays, MHz | replay, random features | replay, contiguous copies of the features | data: zeros % / denormal % of x1 |')
print('|---|---|---:|---:|---|---:|---:|---:|---:|---:|---:|---:|---|')
for idx, x1, w, c_out, n_out, kw, out, ck in captured:
    ev0, ev1, info = trace[idx]
    t_step = ev0.elapsed_time(ev1) * 1e3
    kw2 = dict(kw)
    kw2['out'] = out if kw.get('out') is None else kw['out']
    t_same = queued(lambda: real(x1, w, c_out, n_out, **kw2))
    ck_replay = last_clock[0]
    r1 = torch.randn_like(x1)
    kw3 = dict(kw2)
    if kw.get('x2') is not None:
        kw3['x2'] = torch.randn_like(kw['x2'])
    t_rand = queued(lambda: real(r1, w, c_out, n_out, **kw3))
    kw4 = dict(kw2)
    c1 = x1.contiguous().clone()
    if kw.get('x2') is not None:
        kw4['x2'] = kw['x2'].contiguous().clone()
    t_copy = queued(lambda: real(c1, w, c_out, n_out, **kw4))
    zeros = float((x1 == 0).float().mean()) * 100
    den = float(((x1 != 0) & (x1.abs() < 1.1754944e-38)).float().mean()) * 100
    x2 = kw.get('x2')
    print(f'| {idx} | {info["c_in"]} -> {c_out} | {n_out} | {x1.stride(0)} | {"-" if x2 is None else tuple(x2.shape)} | {kw.get("act", 0)} | {t_step:.1f} | {mhz(ck) if ck >= 0 else 0:.0f} | '
          f'{t_same:.1f} | {mhz(ck_replay):.0f} | {t_rand:.1f} | {t_copy:.1f} | {zeros:.1f} / {den:.2f} |')
