"""The fused rate kernel (fpcc_deep_factorized_bits_f32) against the tensor-op formulation of the same density
(fastpcc_amd.entropy_models._NoisyDeepFactorized.log_prob, itself pinned to the reference by tests/golden/entropy_model.json):
value, gradient w.r.t. the latent and w.r.t. every parameter."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def _model(c, seed, scaler=1.0):
    from fastpcc_amd.entropy_models import NoisyDeepFactorizedEntropyModel
    torch.manual_seed(seed)
    em = NoisyDeepFactorizedEntropyModel(batch_shape=torch.Size([c]), coding_ndim=2, bottleneck_scaler=scaler).cuda()
    with torch.no_grad():                      # move away from the symmetric initialisation
        for p in list(em.prior_weights) + list(em.prior_biases) + list(em.prior_factors):
            p.add_(torch.randn_like(p) * 0.3)
    return em


@pytest.mark.parametrize('n,c,spread', [(1, 1, 1.0), (37, 8, 2.0), (5000, 8, 6.0), (70001, 3, 30.0), (300, 130, 4.0)])
def test_bits_and_gradients(n, c, spread):
    em = _model(c, seed=n + c)
    base = em.prior.base
    params = list(em.prior_weights) + list(em.prior_biases) + list(em.prior_factors)
    y = (torch.randn(n, c, device='cuda') * spread).requires_grad_()

    want = base.log_prob(y).sum()
    want_grads = torch.autograd.grad(want, [y] + params)
    got = base.log_prob_sum(y)
    got_grads = torch.autograd.grad(got, [y] + params)

    assert abs(got.item() - want.item()) <= 2e-5 * abs(want.item()) + 1e-4
    for name, a, b in zip(['y'] + [f'p{i}' for i in range(len(params))], got_grads, want_grads):
        assert a.shape == b.shape, name
        scale = b.abs().max().item() + 1e-6
        # tensor-op sums accumulate in a different order; the elementwise dy is the tight one
        tol = 2e-5 if name == 'y' else 3e-4
        assert (a - b).abs().max().item() <= tol * scale + 1e-6, (name, (a - b).abs().max().item(), scale)


def test_incoming_gradient_scales():
    em = _model(4, seed=3)
    y = torch.randn(100, 4, device='cuda', requires_grad=True)
    (em.prior.base.log_prob_sum(y) * -2.5).backward()
    g1 = y.grad.clone(); w1 = em.prior_weights[2].grad.clone()
    y.grad = None; em.zero_grad()
    (em.prior.base.log_prob(y).sum() * -2.5).backward()
    assert torch.allclose(g1, y.grad, rtol=1e-4, atol=1e-6)
    assert torch.allclose(w1, em.prior_weights[2].grad, rtol=1e-3, atol=1e-5)


def test_training_forward_uses_kernel_and_matches():
    """forward() in training mode: same noise (seeded) -> same bits as the tensor-op path"""
    em = _model(8, seed=5)
    em.train()
    x = torch.randn(1, 2000, 8, device='cuda') * 3
    torch.manual_seed(11)
    y, loss = em(x)
    want = em.prior.log_prob(y).sum() / (-math.log(2))
    assert abs(loss['bits_loss'].item() - want.item()) <= 2e-5 * want.item()
    assert loss['bits_loss'].grad_fn is not None and 'DeepFactorizedBits' in type(loss['bits_loss'].grad_fn).__name__ or \
        'DeepFactorizedBits' in str(loss['bits_loss'].grad_fn.next_functions)


def test_deterministic():
    em = _model(8, seed=7)
    y = torch.randn(30000, 8, device='cuda') * 3
    from fastpcc_amd import hipops as ops
    a = ops.deep_factorized_bits(y, em.prior_weights, em.prior_biases, em.prior_factors, 0.5)
    b = ops.deep_factorized_bits(y, em.prior_weights, em.prior_biases, em.prior_factors, 0.5)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])


def test_rejects_other_networks():
    from fastpcc_amd import hipops as ops
    from fastpcc_amd.hipops import FpccError
    em = _model(8, seed=7)
    y = torch.randn(10, 8, device='cuda')
    with pytest.raises(ValueError):
        ops.deep_factorized_bits(y, em.prior_weights[:4], em.prior_biases, em.prior_factors, 0.5)
    with pytest.raises((FpccError, ValueError)):
        ops.deep_factorized_bits(y, [w.cpu() for w in em.prior_weights], em.prior_biases, em.prior_factors, 0.5)


# ---- scale-indexed noisy normal (Gaussian-conditional) ---------------------------------------------------------------
def _indexed(**kw):
    from fastpcc_amd.entropy_models_indexed import ContinuousIndexedEntropyModel, NoisyNormal, \
        noisy_scale_normal_indexed_entropy_model_init
    return ContinuousIndexedEntropyModel(NoisyNormal, (64,), noisy_scale_normal_indexed_entropy_model_init(0.11, 256, 64),
                                         coding_ndim=2, **kw).cuda()


@pytest.mark.parametrize('n,spread', [(1, 1.0), (1000, 1.0), (70001, 3.0), (5000, 12.0)])
def test_noisy_normal_rate_kernel_against_tensor_ops(n, spread):
    """values from well inside the density to far in its tails (|y| up to ~12 scales: every segment of log Phi)"""
    from fastpcc_amd.entropy_models_indexed import NoisyNormal, _NoisyNormalBits
    a, b = math.log(0.11), math.log(256 / 0.11) / 63
    torch.manual_seed(n)
    idx = (torch.rand(n, device='cuda') * 63).requires_grad_()
    y = (torch.randn(n, device='cuda') * torch.exp(a + b * idx.detach()) * spread).requires_grad_()
    want = NoisyNormal(0, torch.exp(a + b * idx)).log_prob(y).sum()
    wy, wi = torch.autograd.grad(want, [y, idx])
    got = _NoisyNormalBits.apply(y, idx, a, b)
    gy, gi = torch.autograd.grad(got, [y, idx])
    assert torch.isfinite(got) and abs(got.item() - want.item()) <= 3e-5 * abs(want.item()) + 1e-4
    for name, g, w in (('dy', gy, wy), ('di', gi, wi)):
        assert torch.isfinite(g).all(), name
        err = (g - w).abs()
        assert (err <= 2e-3 * w.abs() + 2e-4 * w.abs().max()).all(), (name, err.max().item(), w.abs().max().item())


def test_indexed_model_training_forward_uses_the_kernel():
    em = _indexed(bottleneck_process='')
    em.train()
    x = (torch.randn(1, 3000, 4, device='cuda') * 5).requires_grad_()
    idx = (torch.rand(1, 3000, 4, device='cuda') * 70 - 3).requires_grad_()          # some outside [0, 63]: bounded first
    y, loss = em(x, idx)
    assert 'NoisyNormalBits' in type(loss['bits_loss'].grad_fn).__name__ or 'NoisyNormalBits' in str(loss['bits_loss'].grad_fn.next_functions)
    gx, gi = torch.autograd.grad(loss['bits_loss'], [x, idx])
    bounded = em.bound_indexes(idx)
    want = em.make_prior(bounded).log_prob(x).sum() / (-math.log(2))
    wx, wi = torch.autograd.grad(want, [x, idx])
    assert abs(loss['bits_loss'].item() - want.item()) <= 3e-5 * want.item()
    for g, w in ((gx, wx), (gi, wi)):          # far tails: the autograd of the segment approximations vs analytic derivatives
        assert ((g - w).abs() <= 2e-3 * w.abs() + 2e-4 * w.abs().max()).all()
    outside = (idx < 0) | (idx > 63)
    assert ((gi[outside] == 0) == (wi[outside] == 0)).all()                          # the bound's gradient rule applies to both


def test_indexed_model_codes_on_the_gpu():
    em = _indexed().eval()
    torch.manual_seed(0)
    idx = torch.rand(1, 2000, 8, device='cuda') * 63
    x = torch.randn(1, 2000, 8, device='cuda') * torch.exp(math.log(0.11) + math.log(256 / 0.11) / 63 * idx) * 0.5
    rec, strings = em(x, idx)
    assert torch.equal(rec, x.round().clamp(-64 - 1e9, 64 + 1e9)) or (rec - x).abs().max() <= 0.5 + 1e-6
    assert sum(len(s) for s in strings) > 0


def test_rate_kernels_on_empty_and_tiny_inputs():
    from fastpcc_amd import hipops as ops
    em = _model(4, seed=1)
    out, dy = ops.deep_factorized_bits(torch.empty(0, 4, device='cuda'), em.prior_weights, em.prior_biases, em.prior_factors, 0.5)
    assert out.shape == (4, 59) and float(out.abs().sum()) == 0 and dy.shape == (0, 4)
    total, dy, di = ops.noisy_normal_bits(torch.empty(0, device='cuda'), torch.empty(0, device='cuda'), -2.0, 0.1)
    assert float(total) == 0 and dy.numel() == 0 and di.numel() == 0
    total, dy, di = ops.noisy_normal_bits(torch.tensor([0.0], device='cuda'), torch.tensor([0.0], device='cuda'), 0.0, 0.0)
    want = math.log(math.erf(0.5 / math.sqrt(2)))                      # P(|N(0,1)| < 0.5)
    assert abs(float(total) - want) < 1e-6 and abs(float(dy[0])) < 1e-6
