"""The float LiDAR model (codecs/lossl_coord: trained, calibrated, converted into lossl_coord_int) against
tests/golden/codec_lossl.json -- the REFERENCE's models/convolutional/lossl_coord executed by make_golden.py over the
functional torchsparse stand-in (CPU): module tree, test-time streams, and the loss terms of train_forward; then its own
training path (gradients, AdamW)."""
import json
import os

import numpy as np
import pytest
import torch

from fastpcc_amd.synthetic import batched

pytestmark = pytest.mark.gpu

with open(os.path.join(os.path.dirname(__file__), 'golden', 'codec_lossl.json')) as f:
    G = json.load(f)


def _model(run, train=False):
    from fastpcc_amd.codecs.lossl_coord import Config, Model
    from fastpcc_amd.codecs.lossy_coord_v3.init_random import randomize_
    model = Model(Config(**run['config']), 'cuda')
    randomize_(model, run['seed'])
    model = model.cuda()
    return model.train() if train else model.eval()


@pytest.mark.parametrize('run', G['runs'], ids=[r['label'] for r in G['runs']])
def test_streams_against_the_reference_run(run):
    model = _model(run)
    assert [[k, list(v.shape)] for k, v in model.state_dict().items()] == run['state_dict_keys']
    assert float(sum(p.detach().double().abs().sum() for p in model.parameters())) == pytest.approx(run['param_abs_sum'], rel=1e-12)
    xyz = np.array(run['xyz'], dtype=np.int32)
    data = model.compress(torch.from_numpy(batched(xyz)).cuda())
    want = bytes.fromhex(run['stream_hex'])
    assert data[:8] == want[:8]                                            # offsets and the coarsest level's point count
    assert abs(len(data) - len(want)) <= 0.02 * len(want) + 4             # CDF rows differ by fp32 rounding at most
    rec = model.decompress(data).cpu().numpy()
    assert sorted(map(tuple, rec.tolist())) == sorted(map(tuple, xyz.tolist()))


@pytest.mark.parametrize('run', G['train'], ids=[r['label'] for r in G['train']])
def test_objective_equals_the_reference(run):
    model = _model(run, train=True)
    out = model.train_forward(torch.tensor(run['xyz'], dtype=torch.int32).cuda(), run['points_num'], 0)
    assert set(out) - {'loss'} == set(run['terms'])
    for k, want in run['terms'].items():
        assert out[k] == pytest.approx(want, rel=2e-3, abs=2e-5), k
    assert out['loss'].item() == pytest.approx(run['loss'], rel=2e-3)


def test_gradients_and_adamw():
    run = G['train'][0]
    model = _model(run, train=True)
    xyz, points_num = torch.tensor(run['xyz'], dtype=torch.int32).cuda(), run['points_num']
    loss = model.train_forward(xyz, points_num)['loss']
    loss.backward()
    missing = [n for n, p in model.named_parameters() if p.grad is None or not torch.isfinite(p.grad).all() or p.grad.abs().sum() == 0]
    assert not missing, missing[:8]
    # derivative along the gradient of single tensors (incl. the 4x4x4 embedding of the deepest multi-step predictor)
    named = dict(model.named_parameters())
    embed = [n for n in named if '.embed.0.kernel' in n]
    picks = embed + ['block_dec_recurrent.dec.conv.kernel', 'blocks_dec.0.pred.0.0.kernel', 'blocks_dec.0.pred.3.2.weight', 'blocks_dec.5.upsample.3.weight']

    def value():
        with torch.no_grad():
            return model.train_forward(xyz, points_num)['loss'].item()
    for name in picks:
        prm = named[name]
        g = prm.grad.clone()
        norm = float(g.norm())
        d = g / norm
        eps = min(5e-3 / norm, 0.01 * float(prm.detach().norm()))
        with torch.no_grad():
            prm.add_(eps * d)
            up = value()
            prm.sub_(2 * eps * d)
            down = value()
            prm.add_(eps * d)
        numeric = (up - down) / (2 * eps)
        assert abs(numeric - norm) <= 0.1 * norm + 2e-4 / eps, (name, numeric, norm, eps)
    opt = torch.optim.AdamW(model.parameters(), lr=2e-3)
    values = []
    for _ in range(10):
        loss = model.train_forward(xyz, points_num)['loss']
        opt.zero_grad(set_to_none=True)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
        opt.step()
        values.append(loss.item())
    assert all(np.isfinite(values)) and all(b < a for a, b in zip(values, values[1:])) and values[-1] < values[0] - 1.0, values
