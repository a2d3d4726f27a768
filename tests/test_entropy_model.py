"""NoisyDeepFactorizedEntropyModel (fastpcc_amd.entropy_models) against values produced by the reference's own module
(tests/golden/entropy_model.json, written by tests/golden/make_golden.py): densities, training loss and gradients, the
quantised CDF tables, the compressed strings, state-dict keys, and the gradient-shaping helpers."""
import json
import os

import numpy as np
import pytest
import torch

from fastpcc_amd import entropy_models as EM

G = json.load(open(os.path.join(os.path.dirname(__file__), 'golden', 'entropy_model.json')))


def _model(rec):
    torch.manual_seed(0)
    em = EM.NoisyDeepFactorizedEntropyModel(
        batch_shape=torch.Size([rec['ch']]), coding_ndim=2, bottleneck_process='noise', bottleneck_scaler=rec['scaler'],
        init_scale=rec['init_scale'], lower_bound=rec['lower'], upper_bound=rec['upper'], broadcast_shape_bytes=(3,))
    with torch.no_grad():
        for name, plist in (('weights', em.prior_weights), ('biases', em.prior_biases), ('factors', em.prior_factors)):
            assert len(plist) == len(rec[name])
            for p, v in zip(plist, rec[name]):
                p.copy_(torch.tensor(v, dtype=torch.float64).reshape(p.shape))
    return em


def _t(v, shape=None):
    t = torch.tensor(v, dtype=torch.float64).float()
    return t.reshape(shape) if shape is not None else t


@pytest.mark.parametrize('ci', range(len(G['cases'])))
def test_density_matches_reference(ci):
    rec = G['cases'][ci]
    em = _model(rec)
    probe = _t(rec['probe'], rec['probe_shape'])
    with torch.no_grad():
        lc = EM.logits_cdf(probe, em.prior.batch_shape, em.prior_weights, em.prior_biases, em.prior_factors)
        assert torch.equal(lc.reshape(-1), _t(rec['logits_cdf']))
        assert torch.equal(em.prior.log_prob(probe).reshape(-1), _t(rec['log_prob']))
        assert torch.equal(em.prior.prob(probe).reshape(-1), _t(rec['prob']))


@pytest.mark.parametrize('ci', range(len(G['cases'])))
def test_training_forward_and_gradients(ci):
    rec = G['cases'][ci]
    em = _model(rec).train()
    x = _t(rec['x'], rec['x_shape'])
    torch.manual_seed(rec['train_noise_seed'])
    y, loss = em(x)
    assert torch.equal(y.reshape(-1), _t(rec['train_y']))
    assert float(loss['bits_loss'].detach()) == pytest.approx(rec['bits_loss'], rel=1e-6)
    loss['bits_loss'].backward()
    for got, key in ((em.prior_weights[0].grad, 'grad_w0'), (em.prior_factors[0].grad, 'grad_f0'), (em.prior_biases[-1].grad, 'grad_b_last')):
        want = _t(rec[key])      # tolerance: the fp32 reduction order of autograd sums (terms of mixed sign) is not the reference's
        torch.testing.assert_close(got.reshape(-1), want, rtol=2e-4, atol=2e-4 * float(want.abs().max()))


@pytest.mark.parametrize('ci', range(len(G['cases'])))
def test_cdf_table_and_strings(ci):
    rec = G['cases'][ci]
    em = _model(rec).eval()
    assert [list(map(int, c)) for c in em.prior.cdf_list] == rec['cdfs']
    assert [int(v) for v in em.prior.cdf_offset_list] == rec['cdf_offsets']
    x = _t(rec['x'], rec['x_shape'])
    strings, bshape, deq, bits = em.compress(x.clone(), estimate_bits=True)
    assert [s.hex() for s in strings] == rec['strings']
    assert torch.equal(deq.reshape(-1), _t(rec['dequantized']))
    assert float(bits) == pytest.approx(rec['est_bits'], rel=1e-6)
    back = em.decompress(strings, bshape, torch.device('cpu'))
    assert torch.equal(back, deq)
    out, strings2, _ = em(x.clone())
    assert strings2 == strings and torch.equal(out, deq)
    assert sorted(em.state_dict().keys()) == rec['state_keys']


def test_state_dict_round_trip_carries_the_table():
    rec = G['cases'][1]
    a = _model(rec).eval()
    sd = a.state_dict()
    b = EM.NoisyDeepFactorizedEntropyModel(torch.Size([rec['ch']]), 2, broadcast_shape_bytes=(3,))
    b.load_state_dict(sd)
    assert b.prior.requires_updating_cdf_table is False
    assert b.prior.range_coder.get_cdfs() == a.prior.range_coder.get_cdfs()
    x = _t(rec['x'], rec['x_shape'])
    assert b.compress(x.clone())[0] == a.compress(x.clone())[0]
    # a training-mode state dict flags the table stale and the loader keeps its own
    a.train()
    c = EM.NoisyDeepFactorizedEntropyModel(torch.Size([rec['ch']]), 2, broadcast_shape_bytes=(3,))
    c.load_state_dict(a.state_dict())
    assert c.prior.requires_updating_cdf_table is True


def test_fresh_initialisation_matches_reference():
    torch.manual_seed(0)
    em = EM.NoisyDeepFactorizedEntropyModel(batch_shape=torch.Size([1]), coding_ndim=2, init_scale=10, broadcast_shape_bytes=(3,))
    for p, v in zip(em.prior_biases, G['fresh']['biases']):
        assert torch.equal(p.reshape(-1), _t(v))
    for p, v in zip(em.prior_weights, G['fresh']['weights']):
        assert torch.equal(p.reshape(-1), _t(v))
    assert all(float(f.detach().abs().sum()) == 0 for f in em.prior_factors)
    # 100 values ~ N(0, 3^2) under the fresh prior
    x = torch.randn(1, 100, 1) * 3
    em.eval()
    strings, _, _, bits = em.compress(x.clone(), estimate_bits=True)
    f = G['fresh']
    assert float(bits) == pytest.approx(f['est_bits'], rel=1e-6)
    assert len(em.prior.cdf_list[0]) == f['cdf_len'] and int(em.prior.cdf_offset_list[0]) == f['cdf_offset']
    assert strings[0].hex() == f['string']


@pytest.mark.parametrize('rec', G['bounds'], ids=lambda r: f"{r['fn']}-{r['mode']}")
def test_gradient_shaping(rec):
    x = _t(rec['x']).requires_grad_()
    if rec['fn'] == 'grad_scaler':
        y = EM.grad_scaler(x, float(rec['mode']))
    else:
        y = getattr(EM, rec['fn'] + '_bound')(x, 0.25, rec['mode'])
    y.backward(_t(rec['g']))
    assert torch.equal(y.detach(), _t(rec['y'])) and torch.equal(x.grad, _t(rec['dx']))


def test_bad_arguments():
    with pytest.raises(ValueError):
        EM.NoisyDeepFactorizedEntropyModel(torch.Size([4]), 2, bottleneck_process='banana')
    with pytest.raises(ValueError):
        EM.make_parameters(4, 10, (2, 3, 1))
    with pytest.raises(ValueError):
        EM.NoisyDeepFactorizedEntropyModel(torch.Size([4]), 2, lower_bound=3, upper_bound=3)


def test_sparse_tensor_adapters_without_gpu():
    """minkowski_tensor_wrapped_fn / _op on plain tensors and on objects that only look like sparse tensors are identities
    (the SparseTensor round trip itself needs the device and is in tests/test_gpu_codec_v2.py)"""
    from fastpcc_amd.sparse_conv_layers import get_minkowski_tensor_coords_tuple, minkowski_tensor_wrapped_fn, \
        minkowski_tensor_wrapped_op

    @minkowski_tensor_wrapped_fn({1: 0, '<del>tag': None})
    def f(self, x, scale=2.0, **kw):
        assert 'tag' not in kw
        return x * scale, 'aux'
    y, aux = f(None, torch.ones(3), tag='dropped')
    assert torch.equal(y, torch.full((3,), 2.0)) and aux == 'aux'
    assert torch.equal(minkowski_tensor_wrapped_op(torch.ones(2), lambda t: t + 1), torch.full((2,), 2.0))
    assert get_minkowski_tensor_coords_tuple(torch.ones(1)) is None
