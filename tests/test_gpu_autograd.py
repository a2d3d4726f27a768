"""Back-propagation through the sparse convolutions (fastpcc_amd/autograd.py: fpcc_conv_f32 on mirrored maps for the
input gradient, fpcc_conv_wgrad_f32 for the weight gradient) against a plain PyTorch float64 evaluation of the same
operator built from index_select / matmul / index_add (autograd of torch itself).  Tolerance: 2e-4 of the tensor's
magnitude (fp32 accumulation over up to ~10^5 rows vs float64)."""
import numpy as np
import pytest
import torch

from oracle import coords as oc
from util import batched, surface_cloud

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def scene():
    xyz = surface_cloud(41, 64, 12000)
    lvl = oc.Level(batched(xyz), 1)
    up = oc.strided(lvl)
    k3 = oc.dense_table(oc.kernel_map(lvl, lvl, 3), lvl.n)                 # [27][n]   input row per (offset, output row)
    k2 = oc.dense_table(oc.kernel_map(lvl, up, 2), up.n)                   # [8][m]    child row per (octant, parent)
    return {'n': lvl.n, 'm': up.n, 'k3': torch.from_numpy(k3).cuda(), 'child_row': torch.from_numpy(k2.T.copy()).cuda()}


def _ref_conv(x, w, table_in, n_out, out_rows=None):
    """y[o] = sum_k x[table_in[k][o]] @ w[k]   (table -1 = absent); optional scatter of row o to out_rows[k][o]"""
    y = None
    for k in range(table_in.shape[0]):
        idx = table_in[k].long()
        ok = idx >= 0
        contrib = torch.zeros((n_out, w.shape[-1]), dtype=x.dtype, device=x.device)
        contrib[ok] = x[idx[ok]] @ w[k]
        y = contrib if y is None else y + contrib
    return y


def _close(a, b, what):
    scale = float(b.abs().max()) + 1e-30
    err = float((a.double() - b).abs().max())
    assert err <= 2e-4 * scale, f'{what}: max err {err:.3e} vs magnitude {scale:.3e}'


def _check(kind, scene, c_in, c_out, seed):
    from fastpcc_amd.autograd import ConvSpec, sparse_conv
    g = torch.Generator().manual_seed(seed)
    n, m, cr = scene['n'], scene['m'], scene['child_row']
    ident_n = torch.arange(n, device='cuda', dtype=torch.int32)[None]
    if kind == 'k1':
        spec, n_in, n_out, kk = ConvSpec('k1', n, n), n, n, 1
    elif kind == 'k3':
        spec, n_in, n_out, kk = ConvSpec('k3', n, n, scene['k3']), n, n, 27
    elif kind == 'k2s2':
        spec, n_in, n_out, kk = ConvSpec('k2s2', n, m, cr), n, m, 8
    elif kind == 'k2s2T':
        spec, n_in, n_out, kk = ConvSpec('k2s2T', m, n, cr), m, n, 8
    else:
        spec, n_in, n_out, kk = ConvSpec('gen', m, 8 * m), m, 8 * m, 8
    x = (torch.randn((n_in, c_in), generator=g)).cuda().requires_grad_()
    w = (torch.randn((kk, c_in, c_out), generator=g) / (c_in * max(kk // 2, 1)) ** 0.5).cuda().requires_grad_()
    gy = torch.randn((n_out, c_out), generator=g).cuda()
    y = sparse_conv(x, w if kk > 1 else w[0], spec)
    y.backward(gy)
    xd, wd = x.detach().double().requires_grad_(), w.detach().double().requires_grad_()
    if kind == 'k1':
        yr = xd @ wd[0]
    elif kind == 'k3':
        yr = _ref_conv(xd, wd, scene['k3'], n)
    elif kind == 'k2s2':
        yr = _ref_conv(xd, wd, cr.t(), m)
    else:                                       # parents -> children: y[child(p, g)] = x[p] @ w[g]
        yr = torch.zeros((n_out, c_out), dtype=torch.float64, device='cuda')
        for oct_ in range(8):
            dst = cr[:, oct_].long() if kind == 'k2s2T' else torch.arange(m, device='cuda') * 8 + oct_
            ok = dst >= 0
            yr = yr.index_add(0, dst[ok], xd[ok] @ wd[oct_])
    yr.backward(gy.double())
    _close(y.detach(), yr.detach(), f'{kind} forward')
    _close(x.grad, xd.grad, f'{kind} dX')
    _close(w.grad, wd.grad, f'{kind} dW')


@pytest.mark.parametrize('kind', ['k1', 'k3', 'k2s2', 'k2s2T', 'gen'])
@pytest.mark.parametrize('c_in,c_out', [(128, 128), (256, 128), (128, 64), (64, 32), (16, 64), (128, 1), (32, 1), (1, 16), (1, 64), (16, 8), (8, 1), (32, 8), (64, 16)])
def test_conv_gradients_match_torch_float64(scene, kind, c_in, c_out):
    _check(kind, scene, c_in, c_out, seed=c_in * 7 + c_out)


def test_weight_gradient_accumulates_and_is_reproducible(scene):
    from fastpcc_amd import hipops as ops
    n = scene['n']
    g = torch.Generator().manual_seed(0)
    x = torch.randn((n, 64), generator=g).cuda()
    dy = torch.randn((n, 128), generator=g).cuda()
    a = ops.conv_wgrad(x, dy, n, nbr=scene['k3'], n_offsets=27, nbr_ks=n, nbr_os=1)
    b = ops.conv_wgrad(x, dy, n, nbr=scene['k3'], n_offsets=27, nbr_ks=n, nbr_os=1)
    assert torch.equal(a, b)
    c = a.clone()
    ops.conv_wgrad(x, dy, n, nbr=scene['k3'], n_offsets=27, nbr_ks=n, nbr_os=1, out=c, accumulate=True)
    torch.testing.assert_close(c, 2 * a, rtol=1e-5, atol=1e-4 * float(a.abs().max()))   # (a + p0) + p1 ... vs 2 (p0 + p1 ...)


def test_engine_modules_backpropagate(scene):
    """the ME-named modules in training mode: conv -> PReLU -> strided conv -> transposed conv back -> linear, loss.backward()
    fills every parameter's .grad, and a finite-difference probe of one weight agrees"""
    from fastpcc_amd import engine as ME
    from fastpcc_amd.sparse_conv_layers import ConvBlock, ConvTransBlock, MEMLPBlock
    torch.manual_seed(0)
    xyz = surface_cloud(41, 64, 12000)
    coords = torch.from_numpy(batched(xyz)).to(torch.int32).cuda()
    cm = ME.CoordinateManager(D=3)
    x = ME.SparseTensor(torch.ones((coords.shape[0], 1), device='cuda'), coordinates=coords, coordinate_manager=cm)
    net = torch.nn.ModuleList([ConvBlock(1, 16, 3, 1, act='prelu'), ConvBlock(16, 32, 2, 2, act='prelu'),
                               ConvBlock(32, 32, 3, 1, act='prelu'), ConvTransBlock(32, 16, 2, 2, act='prelu'),
                               MEMLPBlock(16, 1, act=None)]).cuda()

    def loss_of():
        h = net[0](x)
        h = net[1](h)
        h = net[2](h)
        h = net[3](h, x.coordinate_map_key)
        h = net[4](h)
        return (h.F ** 2).mean()
    loss = loss_of()
    loss.backward()
    for name, p in net.named_parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all() and float(p.grad.abs().sum()) > 0, name
    p = net[2].conv.kernel
    idx = (13, 5, 7)
    g_analytic = float(p.grad[idx])
    with torch.no_grad():
        eps = 1e-2
        p[idx] += eps
        up = float(loss_of_nograd(net, x))
        p[idx] -= 2 * eps
        dn = float(loss_of_nograd(net, x))
        p[idx] += eps
    assert (up - dn) / (2 * eps) == pytest.approx(g_analytic, rel=5e-2, abs=1e-6)


def loss_of_nograd(net, x):
    with torch.no_grad():                       # the fused inference path evaluates the same function
        h = net[0](x)
        h = net[1](h)
        h = net[2](h)
        h = net[3](h, x.coordinate_map_key)
        h = net[4](h)
        return (h.F ** 2).mean()


@pytest.mark.parametrize('c', [1, 8, 64, 128, 300])
@pytest.mark.parametrize('act', ['none', 'relu', 'prelu'])
def test_fused_epilogue_backward_matches_torch(c, act):
    """fpcc_epilogue_bwd_f32 (dL/dpre, dbias, dslope from the layer OUTPUT) against torch autograd of bias add + (P)ReLU"""
    from fastpcc_amd import hipops as ops
    g = torch.Generator().manual_seed(c)
    n = 5000 + c
    pre = torch.randn((n, c), generator=g).cuda().double().requires_grad_()
    bias = torch.randn(c, generator=g).cuda().double().requires_grad_()
    slope = torch.tensor([0.3], device='cuda', dtype=torch.float64, requires_grad=True)
    z = pre + bias
    y = z if act == 'none' else torch.relu(z) if act == 'relu' else torch.nn.functional.prelu(z, slope)
    dy = torch.randn((n, c), generator=g).cuda()
    y.backward(dy.double())
    kind = {'none': ops.ACT_NONE, 'relu': ops.ACT_RELU, 'prelu': ops.ACT_PRELU}[act]
    got_g, got_b, got_s = ops.epilogue_bwd(y.detach().float().contiguous(), dy, kind, slope.detach().float() if act == 'prelu' else None,
                                           True, act == 'prelu')
    _close(got_g, pre.grad, 'g')
    _close(got_b, bias.grad, 'dbias')
    if act == 'prelu':
        _close(got_s, slope.grad, 'dslope')
    again = ops.epilogue_bwd(y.detach().float().contiguous(), dy, kind, slope.detach().float() if act == 'prelu' else None, True, False)
    assert torch.equal(again[0], got_g) and torch.equal(again[1], got_b)


@pytest.mark.parametrize('c_in,c_out', [(128, 128), (64, 128), (128, 64), (192, 64)])
def test_weight_gradient_in_pattern_row_order(c_in, c_out):
    """fpcc_conv_wgrad_f32 with a row order: blocks of 32 rows without the offset are skipped, row splits interleave --
    the same sums as without it (another association) and as a float64 evaluation"""
    from fastpcc_amd import hipops as ops
    xyz = surface_cloud(41, 128, 60000)
    lvl = oc.Level(batched(xyz), 1)
    n = lvl.n - 13                                                   # ragged last block
    table = oc.dense_table(oc.kernel_map(lvl, lvl, 3), lvl.n)[:, :n].copy()
    table[table >= n] = -1                                           # inputs restricted to the first n rows too
    nbr = torch.from_numpy(table).cuda()
    order = ops.conv_row_order(nbr, 27, n, 1, n, 13)
    g = torch.Generator().manual_seed(c_in + c_out)
    x = torch.randn(n, c_in, generator=g).cuda()
    dy = torch.randn(n, c_out, generator=g).cuda()
    plain = ops.conv_wgrad(x, dy, n, nbr=nbr, n_offsets=27, nbr_ks=n, nbr_os=1)
    got = ops.conv_wgrad(x, dy, n, nbr=nbr, n_offsets=27, nbr_ks=n, nbr_os=1, row_order=order)
    again = ops.conv_wgrad(x, dy, n, nbr=nbr, n_offsets=27, nbr_ks=n, nbr_os=1, row_order=order)
    assert torch.equal(got, again)                                   # fixed association
    want = torch.zeros(27, c_in, c_out, dtype=torch.float64, device='cuda')
    xd, dd = x.double(), dy.double()
    for k in range(27):
        rows = (nbr[k] >= 0).nonzero().flatten()
        want[k] = xd[nbr[k][rows].long()].t() @ dd[rows]
    scale = want.abs().max().item()
    assert (got.view(27, c_in, c_out).double() - want).abs().max().item() <= 2e-5 * scale
    assert (plain.view(27, c_in, c_out).double() - want).abs().max().item() <= 2e-5 * scale
    # accumulate into an existing gradient
    base = torch.ones_like(got)
    ops.conv_wgrad(x, dy, n, nbr=nbr, n_offsets=27, nbr_ks=n, nbr_os=1, row_order=order, out=base, accumulate=True)
    assert torch.allclose(base, got + 1, rtol=0, atol=1e-4 * scale)


@pytest.mark.parametrize('k,c_in,c_out', [(64, 32, 32), (64, 1, 16), (30, 32, 64), (27, 32, 32)])
def test_general_table_convolution_with_many_offsets(k, c_in, c_out):
    """a lookup-table convolution with more offsets than one launch takes (the 4x4x4 occupancy embedding has 64; the MFMA path takes 27,
    the VALU path 32 per launch): the partition of hipops.table_conv_chunk is the same in the autograd forward and in the inference
    operator, so both give the same bits, and forward / weight gradient match a float64 evaluation"""
    from fastpcc_amd import hipops as ops
    from fastpcc_amd.autograd import ConvSpec, sparse_conv
    g = torch.Generator().manual_seed(k + c_in)
    n_in, n_out = 700, 500
    table = torch.randint(-1, n_in, (n_out, k), generator=g).to(torch.int32)
    table[torch.rand((n_out, k), generator=g) < 0.4] = -1
    table = table.cuda()
    x = torch.randn((n_in, c_in), generator=g).cuda()
    w = (torch.randn((k, c_in, c_out), generator=g) / (k * c_in) ** 0.5).cuda().requires_grad_()
    y = sparse_conv(x, w, ConvSpec('tab', n_in, n_out, table))
    gy = torch.randn((n_out, c_out), generator=g).cuda()
    y.backward(gy)
    xd, wd = x.double(), w.detach().double().requires_grad_()
    yr = torch.zeros((n_out, c_out), dtype=torch.float64, device='cuda')
    for j in range(k):
        idx = table[:, j].long()
        ok = idx >= 0
        yr = yr + torch.where(ok[:, None], xd[idx.clamp(min=0)] @ wd[j], torch.zeros((), dtype=torch.float64, device='cuda'))
    yr.backward(gy.double())
    _close(y.detach(), yr.detach(), 'tab forward')
    _close(w.grad, wd.grad, 'tab dW')
    # the inference partition: the same launches, added in the same order
    step = ops.table_conv_chunk(c_in, c_out, k)
    assert step <= (27 if ops.conv_order(c_in, 0, c_out, 1, 1, 0) else 32)
    out = None
    with torch.no_grad():
        for a in range(0, k, step):
            b = min(a + step, k)
            part = ops.conv_f32(x, w.detach()[a:b].contiguous(), c_out, n_out, nbr=table[:, a:b].contiguous(), n_offsets=b - a, nbr_ks=1,
                                nbr_os=b - a)
            out = part if out is None else out.add_(part)
    assert torch.equal(out, y.detach())
