"""Training path of lossy_coord_v3 and of the float sparse-conv operators it is built from (torchsparse-style Conv3d over the
hash lookup tables): gradients of the operators against a float64 PyTorch evaluation of the same sums, then the model's
objective: finite, differentiable in every parameter that takes part, consistent with finite differences, decreasing
under AdamW."""
import numpy as np
import pytest
import torch

from fastpcc_amd.synthetic import batched, surface_cloud

pytestmark = pytest.mark.gpu


def _sorted_batch(clouds):
    """[(b, x, y, z)] int32 with every sample Morton ('zyx') sorted and unique, samples in batch order"""
    from fastpcc_amd import hipops as ops
    parts = []
    for b, xyz in enumerate(clouds):
        t = torch.from_numpy(np.unique(xyz, axis=0).astype(np.int32)).cuda()
        t = t - t.amin(0)
        _, perm = ops.sort_keys(ops.morton3d_encode(t, (2, 1, 0)))
        t = t[perm.long()]
        parts.append(torch.cat((torch.full((len(t), 1), b, dtype=torch.int32, device='cuda'), t), 1))
    return torch.cat(parts, 0).contiguous(), [len(p) for p in parts]


def _close(a, b, what, tol=3e-4):
    scale = float(b.abs().max()) + 1e-30
    err = float((a.double() - b).abs().max())
    assert err <= tol * scale, f'{what}: max err {err:.3e} vs magnitude {scale:.3e}'


@pytest.mark.parametrize('c_in,c_out,ks,st', [(8, 16, 3, 1), (16, 16, 3, 1), (32, 32, 3, 1), (16, 1, 3, 1), (32, 1, 3, 1), (16, 8, 3, 1),
                                               (1, 16, 3, 1), (16, 16, 2, 2), (32, 32, 2, 2)])
def test_conv3d_gradients(c_in, c_out, ks, st):
    from fastpcc_amd.int_sparse_conv import Conv3d, SparseTensor, _kernel_table
    xyz, _ = _sorted_batch([surface_cloud(3, 64, 9000)])
    torch.manual_seed(c_in * 100 + c_out)
    conv = Conv3d(c_in, c_out, ks, st, bias=True).cuda()
    x = torch.randn((xyz.shape[0], c_in), device='cuda', requires_grad=True)
    sp = SparseTensor(x, xyz, 1)
    out = conv(sp)
    gy = torch.randn_like(out.F)
    out.F.backward(gy)
    # float64 reference from the same lookup table
    _, table = _kernel_table(xyz, out.C, conv.kernel_size, conv.stride, None)
    table = table[:out.C.shape[0]].long() - 1                              # [n_out, K] input row | -1
    xr = x.detach().double().requires_grad_()
    wr = conv.kernel.detach().double().reshape(conv.kernel_volume, c_in, c_out).requires_grad_()
    br = conv.bias.detach().double().requires_grad_()
    y = torch.zeros((out.C.shape[0], c_out), dtype=torch.float64, device='cuda')
    for k in range(conv.kernel_volume):
        rows = torch.nonzero(table[:, k] >= 0)[:, 0]
        y = y.index_add(0, rows, xr[table[rows, k]] @ wr[k])
    y = y + br
    y.backward(gy.double())
    _close(out.F.detach(), y.detach(), 'forward')
    _close(x.grad, xr.grad, 'dx')
    _close(conv.kernel.grad.reshape(wr.shape), wr.grad, 'dw')
    _close(conv.bias.grad, br.grad, 'dbias')


def _model(seed=0, **kw):
    from fastpcc_amd.codecs.lossy_coord_v3 import Config, Model
    from fastpcc_amd.codecs.lossy_coord_v3.init_random import randomize_
    cfg = Config(channels=32, max_stride=64, warmup_steps=0, coord_recon_loss_factor=2.0, **kw)
    model = Model(cfg)
    randomize_(model, seed)
    return cfg, model.cuda().train()


@pytest.mark.parametrize('kw', [dict(num_latents=(0, 0, 2, 2, 0), lossl_geo_upsample=(0, 1, 1, 1, 1)),        # dense_r1
                                dict(num_latents=(0, 0, 2, 2, 0), lossl_geo_upsample=(0, 0, 1, 1, 1)),        # dense_r4
                                dict(num_latents=(0, 0, 0, 2, 2), lossl_geo_upsample=(0, 0, 0, 1, 1))])       # dense_r7
def test_objective_and_gradients(kw, monkeypatch):
    from fastpcc_amd.codecs.lossy_coord_v3 import model as v3
    cfg, model = _model(1, **kw)
    xyz, points_num = _sorted_batch([surface_cloud(11, 64, 5000), surface_cloud(12, 64, 3500)])
    # the top-k masks of lossy levels are piecewise-constant functions of the parameters (autograd treats them as constants):
    # record them on the first pass and replay them in the finite-difference passes, so that those see the same smooth piece
    masks, real_top_children = [], v3.top_children
    monkeypatch.setattr(v3, 'top_children', lambda logits, n: (masks.append(real_top_children(logits, n)), masks[-1])[1])
    torch.manual_seed(5)
    out = model.train_forward(xyz, points_num, training_step=10)
    loss = out['loss']
    assert torch.isfinite(loss) and loss.item() > 0
    terms = {k: v for k, v in out.items() if k != 'loss'}
    assert abs(sum(terms.values()) - loss.item()) <= 1e-3 * loss.item()
    n_lossy = next((i for i, v in enumerate(cfg.lossl_geo_upsample) if v), len(cfg.lossl_geo_upsample))
    assert {f'stride{2 ** i}_geo_loss' for i in range(1, 7)} <= set(terms)
    assert sum(1 for k in terms if '_fea' in k) == sum(cfg.num_latents)
    loss.backward()
    used = {n for n, p in model.named_parameters() if p.grad is not None and torch.isfinite(p.grad).all() and p.grad.abs().sum() > 0}
    unused = {n for n, _ in model.named_parameters()} - used
    # parameters without a gradient: the upsampling head of the finest predictor does not exist; everything else takes part
    assert not [n for n in unused if 'blocks_dec.0.upsample' not in n], sorted(unused)[:8]

    # Derivative along the gradient of single parameter tensors against central differences of the (noise-seeded)
    # objective.  One tensor at a time and along its own gradient: the objective is fp32 (resolution ~1e-6 of its value)
    # and piecewise smooth (PReLU kinks, top-k masks), so the step must change it by ~1e-2 without leaving the piece.
    named = dict(model.named_parameters())
    lossless = [i for i, v in enumerate(cfg.lossl_geo_upsample) if v]
    picks = [f'blocks_dec.{lossless[0]}.pred.2.weight', 'block_dec_recurrent.dec.conv.kernel', 'block_dec_recurrent.upsample.3.weight',
             'blocks_enc.1.0.kernel', f'blocks_dec.{lossless[0]}.pred.0.kernel']
    latent_level = next(i for i, v in enumerate(cfg.num_latents) if v)
    picks += [f'blocks_dec.{latent_level}.transforms.0.1.4.kernel', f'blocks_dec.{latent_level}.transforms.0.4.prior_biases.1',
              f'blocks_dec.{latent_level}.transforms.0.3.0.weight']
    if n_lossy:
        picks.append('blocks_dec.0.pred.2.kernel')

    recorded = list(masks)

    def value():
        replay = iter(recorded)
        monkeypatch.setattr(v3, 'top_children', lambda logits, n: next(replay))
        with torch.no_grad():
            torch.manual_seed(5)
            return model.train_forward(xyz, points_num, training_step=10)['loss'].item()

    for name in picks:
        prm = named[name]
        g = prm.grad.clone()
        norm = float(g.norm())
        assert norm > 0, name
        d = g / norm
        eps = min(2e-3 / norm, 0.01 * float(prm.detach().norm()))
        with torch.no_grad():
            prm.add_(eps * d)
            up = value()
            prm.sub_(2 * eps * d)
            down = value()
            prm.add_(eps * d)
        numeric = (up - down) / (2 * eps)
        assert abs(numeric - norm) <= 0.12 * norm + 3e-5 / eps, (name, numeric, norm, eps)


def test_adamw_reduces_the_objective():
    cfg, model = _model(2, num_latents=(0, 0, 2, 2, 0), lossl_geo_upsample=(0, 1, 1, 1, 1))
    xyz, points_num = _sorted_batch([surface_cloud(21, 64, 4000), surface_cloud(22, 64, 4000)])
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=1e-4)
    values = []
    for step in range(12):
        torch.manual_seed(step)
        loss = model.train_forward(xyz, points_num, training_step=step)['loss']
        opt.zero_grad(set_to_none=True)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
        opt.step()
        values.append(loss.item())
    assert all(np.isfinite(values)) and min(values[-3:]) < 0.9 * values[0], values
    # the trained model still codes and decodes
    model.eval()
    one, _ = _sorted_batch([surface_cloud(21, 64, 4000)])
    rec = model.decompress(model.compress(one))
    assert rec.shape[1] == 3 and len(rec) > 0


def _train_goldens():
    import json, os
    with open(os.path.join(os.path.dirname(__file__), 'golden', 'codec_v3.json')) as f:
        return json.load(f)['train']


@pytest.mark.parametrize('run', _train_goldens(), ids=[r['label'] for r in _train_goldens()])
def test_objective_equals_the_reference(run, monkeypatch):
    """tests/golden/codec_v3.json['train']: loss terms of the REFERENCE's train_forward (make_golden.py, CPU, functional
    torchsparse stand-in) with the latents' uniform noise replaced by zeros on both sides"""
    from fastpcc_amd.codecs.lossy_coord_v3 import Config, Model
    from fastpcc_amd.codecs.lossy_coord_v3.init_random import randomize_
    cfg = Config(**{k: tuple(v) if isinstance(v, list) else v for k, v in run['config'].items()})
    model = Model(cfg)
    randomize_(model, run['seed'])
    model = model.cuda().train()
    monkeypatch.setattr(torch.Tensor, 'uniform_', lambda self, *a, **k: self.zero_())
    xyz = torch.tensor(run['xyz'], dtype=torch.int32).cuda()
    out = model.train_forward(xyz, run['points_num'], run['training_step'])
    assert set(out) - {'loss'} == set(run['terms'])
    for k, want in run['terms'].items():
        assert out[k] == pytest.approx(want, rel=2e-3, abs=2e-5), k
    assert out['loss'].item() == pytest.approx(run['loss'], rel=2e-3)
