"""Does a 3x3x3 layer slow down under sustained load?  150 back-to-back launches of one layer with per-launch events, and a one-wave
clock probe (fpcc_clock_probe) running on a second stream BESIDE every tenth launch: the shader clock the layer actually gets."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fastpcc_amd import engine as ME, hipops as ops
from fastpcc_amd.synthetic import SCALE, batched, body_cloud

frame = torch.from_numpy(batched(body_cloud(1024, SCALE[1024], seed=2))).cuda()
cm = ME.CoordinateManager(D=3)
x = ME.SparseTensor(torch.ones((frame.shape[0], 1), device='cuda'), coordinates=frame, coordinate_manager=cm)
m = cm._map(x.coordinate_map_key)
maps = []
for _ in range(3):
    m = cm._ensure_parent(m)
    maps.append(m)
side = torch.cuda.Stream()
alive = torch.cuda.Stream()
REPS = int(os.environ.get('REPS', '150'))
KEEP = int(os.environ.get('KEEP', '0'))          # 1: a one-wave spin kernel runs on a third stream from 50 ms before the launches on
keep_out = torch.zeros((64, 2), dtype=torch.int64, device='cuda')
for level, c_in, c_out in ((1, 64, 64), (2, 128, 128)):
    m = maps[level - 1]
    n = m.n
    nbr, order = cm._nbr27(m), cm._row_order(m)
    f = torch.randn((n, c_in), device='cuda')
    w = torch.randn((27, c_in, c_out), device='cuda') / (13 * c_in) ** 0.5
    out = torch.empty((n, c_out), device='cuda')
    fn = lambda: ops.conv_f32(f, w, c_out, n, nbr=nbr, n_offsets=27, nbr_ks=n, nbr_os=1, row_order=order, pack=True, out=out)
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    import time
    if KEEP:
        with torch.cuda.stream(alive):
            for j in range(5 + REPS * 2 // 10):
                ops.clock_probe(keep_out[j % 64], 10000)
    time.sleep(0.05)
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(REPS + 1)]
    clocks = torch.zeros((REPS, 2), dtype=torch.int64, device='cuda')
    evs[0].record()
    for i in range(REPS):
        if i % 10 == 5:
            side.wait_event(evs[i])            # starts beside launch i
            with torch.cuda.stream(side):
                ops.clock_probe(clocks[i], 100)
        fn()
        evs[i + 1].record()
    evs[REPS].synchronize()
    ts = [evs[i].elapsed_time(evs[i + 1]) * 1e3 for i in range(REPS)]
    ck = clocks.tolist()
    print(f'## {c_in} -> {c_out} on {n} rows: {REPS} launches back to back after 50 ms of ' + ('a one-wave spin kernel' if KEEP else 'idle'))
    for i in range(0, REPS, 10):
        c, t = ck[i + 5]
        print(f'launches {i:3d}-{i + 9:3d}: ' + ' '.join(f'{v:7.1f}' for v in ts[i:i + 10]) + f'   clock beside launch {i + 5}: {c / max(t, 1) * 100:6.0f} MHz')
