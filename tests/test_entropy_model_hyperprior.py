"""Hyperprior wrapper (hyper-encoder -> batched deep-factorised side information -> hyper-decoder -> index-conditioned prior)
against the reference's outputs for the same parameters (tests/golden/entropy_model_hyperprior.json)."""
import json
import os

import pytest
import torch
import torch.nn as nn

from fastpcc_amd import entropy_models_hyperprior as H

with open(os.path.join(os.path.dirname(__file__), 'golden', 'entropy_model_hyperprior.json')) as f:
    GOLD = json.load(f)


def _build(case):
    c, ch = case['c'], case['ch']
    if case['name'] == 'scale_normal':
        em = H.ScaleNoisyNormalEntropyModel(nn.Linear(c, ch), nn.Linear(ch, c), torch.Size([ch]), 2, num_scales=32, scale_min=0.2,
                                            scale_max=40, bottleneck_process='')
    else:
        em = H.NoisyDeepFactorizedEntropyModel(nn.Linear(c, ch), nn.Linear(ch, c * 3), torch.Size([ch]), 2, index_ranges=(4, 4, 4),
                                               parameter_fns_type='transform', parameter_fns_factory=lambda i, o: nn.Linear(i, o),
                                               num_filters=(1, 2, 1), bottleneck_process='')
    own = em.state_dict()
    tensors = {k for k, v in own.items() if isinstance(v, torch.Tensor)}
    assert tensors == set(case['state']), sorted(tensors ^ set(case['state']))         # same parameter names as the reference
    state = {k: torch.tensor(v, dtype=torch.float64).float().reshape(shape) for k, (v, shape) in case['state'].items()}
    em.load_state_dict(state, strict=False)
    return em


@pytest.mark.parametrize('case', GOLD, ids=lambda c: c['name'])
def test_hyperprior_model(case):
    em = _build(case)
    y = torch.tensor(case['y'], dtype=torch.float64).float().reshape(1, case['n'], case['c'])
    em.train()
    yg = y.clone().requires_grad_()
    _, loss = em(yg)
    assert set(loss) == {'bits_loss', 'hyper_bits_loss'}
    (loss['bits_loss'] + loss['hyper_bits_loss']).backward()
    for k in ('bits_loss', 'hyper_bits_loss'):
        assert abs(loss[k].item() - case['train'][k]) <= 3e-6 * abs(case['train'][k]), k
    want = torch.tensor(case['train']['dy'], dtype=torch.float64)
    assert (yg.grad.double().flatten() - want).abs().max() <= 3e-5 * want.abs().max()
    em.eval()
    strings, shape, deq, bits = em.compress(y.clone(), estimate_bits=True)
    assert [b.hex() for b in strings] == case['strings']
    assert list(shape) == case['coding_batch_shape']
    assert abs(bits.item() - case['estimated_bits']) <= 3e-6 * case['estimated_bits']
    rec = em.decompress(strings, shape, torch.device('cpu'))
    assert torch.equal(rec.double().flatten(), torch.tensor(case['decoded'], dtype=torch.float64))
    rec2, strings2, shape2 = em(y.clone())
    assert torch.equal(rec2, rec) and strings2 == strings


def test_framing_of_the_two_strings():
    em = _build(GOLD[0])
    framed = em.concat_bytes_lists([b'ab', b''], [b'xyz', b'q'])
    assert framed == [b'\x02\x00abxyz', b'\x00\x00q']
    assert em.split_bytes_lists(framed) == ([b'ab', b''], [b'xyz', b'q'])
    with pytest.raises(ValueError):
        em.split_bytes_lists([b'\x09\x00ab'])
