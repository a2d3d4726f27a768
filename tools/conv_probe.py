#!/usr/bin/env python3
"""One 3x3x3 convolution layer on a pyramid level of the headline cloud, in isolation (for rocprofv3 --pmc passes and
quick A/B timing).  usage: conv_probe.py [level=2] [c_in=128] [c_out=128] [reps=20] [resolution=1024]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
from fastpcc_amd import engine as ME, hipops as ops
from fastpcc_amd.synthetic import SCALE, batched, body_cloud

level = int(sys.argv[1]) if len(sys.argv) > 1 else 2
c_in = int(sys.argv[2]) if len(sys.argv) > 2 else 128
c_out = int(sys.argv[3]) if len(sys.argv) > 3 else 128
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 20
res = int(sys.argv[5]) if len(sys.argv) > 5 else 1024
frame = torch.from_numpy(batched(body_cloud(res, SCALE.get(res, 1.0), seed=2))).cuda()
cm = ME.CoordinateManager(D=3)
x = ME.SparseTensor(torch.ones((frame.shape[0], 1), device='cuda'), coordinates=frame, coordinate_manager=cm)
m = cm._map(x.coordinate_map_key)
for _ in range(level):
    m = cm._ensure_parent(m)
nbr = cm._nbr27(m)
order = cm._row_order(m)
n = m.n
torch.manual_seed(0)
f = torch.randn((n, c_in), device='cuda')
w = torch.randn((27, c_in, c_out), device='cuda') / (13 * c_in) ** 0.5
pairs = int((nbr >= 0).sum().item())
PACK = bool(int(os.environ.get('PACK', '1')))
# ROWS=0: offset-major neighbour table [27, n]; default: the row-major copy the engine hands the MFMA kernels
ROWS = os.environ.get('ROWS', '1') != '0'
def run(row_order):
    return ops.conv_f32(f, w, c_out, n, row_order=row_order, pack=PACK, **cm._k3_table(m, ROWS, row_order if row_order is order else None))
def lpt(order, group):
    present = (nbr >= 0)[:, order.long()]
    pad = (-n) % group
    w = torch.nn.functional.pad(present, (0, pad)).reshape(27, -1, group).any(2).sum(0)          # offsets per group
    full = n // group
    gp = torch.sort(w[:full], descending=True, stable=True)[1]
    body = order[:full * group].reshape(full, group)[gp].reshape(-1)
    return torch.cat([body, order[full * group:]]).contiguous()
cases = [('natural', None), ('pattern', order)]
if os.environ.get('ONLY'):
    cases = [c for c in cases if c[0] == os.environ['ONLY']]
variants = [(None, None)]
DBG = [int(v) for v in os.environ.get('DBG', '0').split(',')]
if os.environ.get('SWEEP'):
    variants = [(0, 0)] + [(nbw, sb) for nbw in (1, 2, 4) if nbw <= c_out // 32 for sb in (0, 1)]
if os.environ.get('NBWS'):
    variants = [(int(v), 1) for v in os.environ['NBWS'].split(',')]
DEPTHS = [int(v) for v in os.environ.get('DEPTHS', '0').split(',')]
if os.environ.get('LPT'):
    cases.append(('pattern+lpt', lpt(order, int(os.environ['LPT']))))
if os.environ.get('W22'):        # W22=1: the default unit against 64 x 64 wave tiles
    variants = [(-1, 1), (-2, 1)]
if os.environ.get('GROUPED'):    # GROUPED=1 (needs FPCC_EXPERIMENT=1): order 1 default unit against the grouped (order 3) evaluation
    variants = [(-1, 1), (-3, 1), (-4, 1)]
if os.environ.get('LDS'):        # LDS=0,2,3,4: the default kernels (0) against k_conv_lds with 2 | 3 | 4 row blocks per workgroup
    variants = [(-100 - int(v), 1) for v in os.environ['LDS'].split(',')]
if os.environ.get('FORMS'):      # FORMS=1: the three forms of the order-3 evaluation -- four waves x 32 columns, four waves x 64 columns, folded
    variants = [(-201, 1), (-202, 1), (-203, 1)]
for dbg, (nbw, sb) in [(a, c) for a in DBG for c in variants]:
  ops.conv_set_tuning(3, dbg)
  tag = ''
  if nbw is not None and nbw <= -201:
      ops.conv_set_tuning(ops.KNOB_GROUPED_FOLD_ROWS, 1 if nbw == -203 else 0)
      ops.conv_set_tuning(ops.KNOB_GROUPED_NBW, {-201: 1, -202: 2, -203: 0}[nbw])
      tag = {-201: ' [grouped, 32 columns per workgroup]', -202: ' [grouped, 64 columns per workgroup]', -203: ' [folded]'}[nbw]
      nbw = None
  if nbw is not None and nbw <= -100:
      rb = -100 - nbw
      ops.conv_set_tuning(ops.KNOB_LDS_ROWS, 1 if rb else 0)
      if rb:
          ops.conv_set_tuning(ops.KNOB_LDS_ROW_BLOCKS, rb)
      tag = f' [lds kernel, {rb} row blocks, dbg={dbg}]' if rb else f' [default kernel dbg={dbg}]'
      nbw = None
  if nbw is not None and nbw < 0:
      ops.conv_set_tuning(ops.KNOB_WAVE_ON, 1); ops.conv_set_tuning(ops.KNOB_WAVE_NBW, 0); ops.conv_set_tuning(ops.KNOB_WAVE_SB, sb)
      ops.conv_set_tuning(ops.KNOB_WAVE22_ROWS, 1 if nbw == -2 else 0)
      tag = ' [wave 64x64]' if nbw == -2 else ' [wave default unit]'
      if os.environ.get('GROUPED'):
          ops.conv_set_tuning(ops.KNOB_GROUPED_OFF, 0 if nbw in (-3, -4) else 1)
          ops.conv_set_tuning(ops.KNOB_GROUPED_NBW, 1 if nbw == -3 else 2)
          tag = {-1: ' [order 1, default unit]', -3: ' [grouped, 32 columns per workgroup]', -4: ' [grouped, 64 columns per workgroup]'}[nbw]
      nbw = None
  if nbw is not None:
      ops.conv_set_tuning(ops.KNOB_WAVE_ON, int(nbw > 0)); ops.conv_set_tuning(ops.KNOB_WAVE_NBW, nbw); ops.conv_set_tuning(ops.KNOB_WAVE_SB, sb)
      tag = f' [tiled kernel dbg={dbg}]' if nbw == 0 else f' [wave nbw={nbw} sb={sb} dbg={dbg}]'
  for name, ro in cases:
    # MI355X's power management starts a burst of matrix work near 2.0 GHz and needs ~30 ms of uninterrupted load to reach 2.4 GHz
    # (profiles/r03/clock_ramp.md): without this warm-up whichever variant runs LAST looks 5-10 % faster than the first
    for _ in range(3):
        run(ro)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    while time.perf_counter() - t0 < float(os.environ.get('WARM_MS', '60')) * 1e-3:
        for _ in range(4):
            run(ro)
        torch.cuda.synchronize()                              # (a few us of idle per four launches do not reset the ramp)
    t0 = time.perf_counter()
    for _ in range(reps):
        run(ro)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
    print(f'level {level} rows {n} pairs/row {pairs / n:.2f} {c_in}->{c_out} {name} order{tag}: {dt * 1e6:.1f} us  '
          f'{2 * pairs * c_in * c_out / dt / 1e12:.1f} TFLOP/s algorithmic')
