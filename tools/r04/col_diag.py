"""where do the colour decoder's outputs leave the chain-order oracle's?  (diagnosis for tests/test_gpu_codec_color.py)"""
import sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from fastpcc_amd.engine import summation_order as ME_order
from oracle.codec_color import OracleColor
from oracle.codec_v2 import Feature
from util import batched, enliven, surface_cloud
from test_gpu_codec_color import _colors
from fastpcc_amd.codecs.lossy_coord_lossy_color import Model
from fastpcc_amd.codecs.lossy_coord_lossy_color.model_config import baseline_r1

cfg = baseline_r1()
torch.manual_seed(0)
model = Model(cfg)
enliven(model, 3, gain=2.3)
weights = {k: v.clone() for k, v in model.state_dict().items() if isinstance(v, torch.Tensor)}
model = model.cuda().eval()
xyz = surface_cloud(7, 64, 16000)
coords = batched(xyz) + np.array([0, 2, 9, 0])
color = _colors(xyz, 1)
perm = np.random.default_rng(0).permutation(len(xyz))
data = model.compress(torch.from_numpy(coords[perm]).to(torch.int32).cuda(), torch.from_numpy(color[perm]).cuda())

cap = {}
def hook(name):
    def f(mod, inp, out):
        cap[name] = (out.C.cpu().numpy().copy(), out.F.detach().cpu().numpy().copy())
    return f
def prehook(mod, inp):
    x = inp[0]
    cap['in'] = (x.C.cpu().numpy().copy(), x.F.detach().cpu().numpy().copy())
pb = model.decoder.predict_block
pb.register_forward_pre_hook(prehook)
for i in range(3):
    pb[i].register_forward_hook(hook(f'l{i}'))
rec_xyz, rec_rgb = model.decompress(data)
rec_xyz, rec_rgb = rec_xyz.cpu().numpy(), rec_rgb.cpu().numpy()

o = OracleColor(weights, cfg, conv='chain', order_fn=ME_order)
want = o.compress(coords, color)
print('bytes equal', data == want, len(data), len(want))
ocap = {}
orig = o.conv_block
def spy(name, fea, kind, **kw):
    if name.startswith('decoder.predict_block'):
        if name.endswith('.0'):
            x2 = kw.get('x2')
            ocap['in'] = (fea.level.coords.copy(), np.concatenate((fea.f.numpy(), x2.f.numpy()), 1))
    out = orig(name, fea, kind, **kw)
    if name.startswith('decoder.predict_block'):
        ocap['l' + name[-1]] = (out.level.coords.copy(), out.f.numpy().copy())
    return out
o.conv_block = spy
o_xyz, o_rgb = o.decompress(data)
print('xyz equal', (o_xyz == rec_xyz).all(), 'rgb rows differing', int((o_rgb != rec_rgb).any(1).sum()), 'of', len(o_rgb),
      'max', float(np.abs(o_rgb - rec_rgb).max()))
d = np.abs(o_rgb - rec_rgb)
print('per channel differing', (d > 0).sum(0), 'hist', np.bincount(np.minimum(d.max(1), 20).astype(int)))
for k in ('in', 'l0', 'l1', 'l2'):
    gc, gf = cap[k]; oc_, of = ocap[k]
    key = lambda c: (c[:, 1].astype(np.int64) << 40) | (c[:, 2].astype(np.int64) << 20) | c[:, 3].astype(np.int64)
    gi, oi = np.argsort(key(gc)), np.argsort(key(oc_))
    same_c = (gc[gi] == oc_[oi]).all() if gc.shape == oc_.shape else False
    if not same_c:
        print(k, 'coordinate sets differ', gc.shape, oc_.shape); continue
    a, b = gf[gi], of[oi]
    bad = (a != b)
    print(k, a.shape, 'coords equal; feature rows differing', int(bad.any(1).sum()), 'max abs', float(np.abs(a - b).max()),
          'channels differing', np.nonzero(bad.any(0))[0][:12], 'nan', int(np.isnan(a).sum()))
