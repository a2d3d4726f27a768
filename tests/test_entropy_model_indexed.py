"""Index-conditioned (Gaussian-conditional) entropy model against values produced by the reference's implementation
(tests/golden/entropy_model_indexed.json; generator make_golden.py:make_entropy_model_indexed)."""
import hashlib
import json
import os

import pytest
import torch

from fastpcc_amd.entropy_models_indexed import ContinuousIndexedEntropyModel, NoisyNormal, log_ndtr, \
    noisy_scale_normal_indexed_entropy_model_init

with open(os.path.join(os.path.dirname(__file__), 'golden', 'entropy_model_indexed.json')) as f:
    GOLD = json.load(f)


def test_log_ndtr_segments():
    g = GOLD['log_ndtr']
    x = torch.tensor(g['x'], dtype=torch.float32)
    got32, want32 = log_ndtr(x).double(), torch.tensor(g['f32'], dtype=torch.float64)
    assert torch.allclose(got32, want32, rtol=2e-6, atol=1e-30)
    got64, want64 = log_ndtr(x.double()), torch.tensor(g['f64'], dtype=torch.float64)
    assert torch.allclose(got64, want64, rtol=1e-12, atol=1e-300)


def test_noisy_normal_density():
    g = GOLD['noisy_normal']
    y = torch.tensor(g['y'], dtype=torch.float32)[:, None]
    scale = torch.tensor(g['scale'], dtype=torch.float32)[None]
    d = NoisyNormal(0, scale)
    assert torch.allclose(d.log_prob(y).double().flatten(), torch.tensor(g['log_prob'], dtype=torch.float64), rtol=3e-6, atol=1e-6)
    assert torch.allclose(d.prob(y).double().flatten(), torch.tensor(g['prob'], dtype=torch.float64), rtol=1e-5, atol=1e-9)


@pytest.mark.parametrize('case', GOLD['models'], ids=lambda c: f"seed{c['seed']}")
def test_scale_indexed_model(case):
    kw = case['kw']
    em = ContinuousIndexedEntropyModel(NoisyNormal, (64,), noisy_scale_normal_indexed_entropy_model_init(0.11, 256, 64),
                                       coding_ndim=2, bottleneck_process='', **kw)
    n, c = case['n'], case['c']
    x = torch.tensor(case['x'], dtype=torch.float64).float().reshape(1, n, c)
    idx = torch.tensor(case['indexes'], dtype=torch.float64).float().reshape(1, n, c)
    assert torch.equal(em.bound_indexes(idx).double().flatten(), torch.tensor(case['bounded'], dtype=torch.float64))
    assert em.flatten_indexes(em.bound_indexes(idx)).flatten().tolist() == case['flat']

    em.train()
    xg, ig = x.clone().requires_grad_(), idx.clone().requires_grad_()
    y, loss = em(xg, ig)
    loss['bits_loss'].backward()
    assert torch.equal(y.detach(), x) == case['train']['y_equals_x']
    assert abs(loss['bits_loss'].item() - case['train']['bits']) <= 2e-6 * abs(case['train']['bits'])
    for got, want in ((xg.grad, case['train']['dx']), (ig.grad, case['train']['di'])):
        want = torch.tensor(want, dtype=torch.float64)
        assert (got.double().flatten() - want).abs().max() <= 2e-5 * want.abs().max() + 1e-9

    em.eval()
    table = [list(map(int, r)) for r in em.prior.cdf_list]
    for i, row in case['table_rows'].items():
        assert table[int(i)] == row, i
    assert hashlib.sha256(json.dumps(table).encode()).hexdigest()[:16] == case['table_sha']
    assert list(map(int, em.prior.cdf_offset_list)) == case['offsets']
    strings, deq, est = em.compress(x.clone(), idx, estimate_bits=True)
    assert [b.hex() for b in strings] == case['strings']
    assert abs(est.item() - case['estimated_bits']) <= 2e-6 * case['estimated_bits']
    rec = em.decompress(strings, idx, torch.device('cpu'))
    assert torch.equal(rec, deq) == case['roundtrip']
    assert torch.equal(rec.double().flatten(), torch.tensor(case['decoded'], dtype=torch.float64))


def test_state_dict_carries_the_table():
    make = lambda: ContinuousIndexedEntropyModel(NoisyNormal, (64,), noisy_scale_normal_indexed_entropy_model_init(0.11, 256, 64),
                                                 coding_ndim=2)
    a = make().eval()
    state = a.state_dict()
    assert 'prior._extra_state' in state and 'range_coding_prior_indexes' not in state
    b = make()
    b.load_state_dict(state)
    assert [list(map(int, r)) for r in b.prior.cdf_list] == [list(map(int, r)) for r in a.prior.cdf_list]
