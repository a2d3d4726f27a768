"""fpcc_conv_f32 (MFMA and VALU paths, through the C ABI) against the oracle.

Bit-exactness: the device kernel documents its summation order (include/fpcc_hip.h); oracle/sparse_conv.c evaluates the
same FMA chain, so outputs are compared BIT FOR BIT.  Against the reference-shaped evaluation (gather, GEMM,
scatter-add) the tolerance is 2e-5 relative to the row's magnitude -- fp32 re-association only."""
import numpy as np
import pytest
import torch

from oracle import coords as oc
from oracle import sparse_conv as sc
from util import batched, surface_cloud

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ops():
    from fastpcc_amd import hipops
    return hipops


@pytest.fixture(scope='module')
def scene():
    xyz = surface_cloud(21, 128, 60000)
    lvl = oc.Level(batched(xyz), 1)
    up = oc.strided(lvl)
    return {'lvl': lvl, 'up': up, 'k3': oc.dense_table(oc.kernel_map(lvl, lvl, 3), lvl.n),
            'k2': oc.dense_table(oc.kernel_map(lvl, up, 2), up.n)}


def _cuda(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


SHAPES_K3 = [(1, 0, 16), (16, 0, 8), (32, 0, 1), (64, 0, 64), (64, 0, 128), (128, 0, 64), (128, 0, 128),
             (128, 128, 128), (128, 0, 1), (16, 0, 64), (48, 0, 32), (8, 8, 4), (5, 0, 3)]


@pytest.mark.parametrize('c1,c2,c_out', SHAPES_K3)
def test_conv3_bit_exact(ops, scene, c1, c2, c_out):
    rng = np.random.default_rng(c1 * 1000 + c2 * 10 + c_out)
    lvl, table = scene['lvl'], scene['k3']
    n = lvl.n
    x1 = rng.normal(size=(n, c1)).astype(np.float32)
    x2 = rng.normal(size=(n, c2)).astype(np.float32) if c2 else None
    w = (rng.normal(size=(27, c1 + c2, c_out)) / np.sqrt(13 * (c1 + c2))).astype(np.float32)
    b = rng.normal(size=c_out).astype(np.float32)
    slope = torch.tensor([0.2], device='cuda')
    got = ops.conv_f32(_cuda(x1), _cuda(w), c_out, n, x2=None if x2 is None else _cuda(x2), nbr=_cuda(table),
                       n_offsets=27, nbr_ks=n, nbr_os=1, bias=_cuda(b), act=ops.ACT_PRELU, slope=slope, clip=1.5)
    order = ops.conv_order(c1, c2, c_out, 27, 1, n)
    want = sc.conv_chain(x1, table, w, b, n, x2=x2, act=sc.ACT_PRELU, slope=0.2, clip=1.5, order=order)
    assert (_bits(got.cpu().numpy()) == _bits(want)).all()
    # reference-shaped evaluation, tolerance
    x = x1 if x2 is None else np.concatenate((x1, x2), 1)
    km = [(table[k][table[k] >= 0].astype(np.int64), np.nonzero(table[k] >= 0)[0]) for k in range(27)]
    ref = sc.conv_mm(torch.from_numpy(x), km, torch.from_numpy(w), torch.from_numpy(b), n, sc.ACT_PRELU, 0.2, 1.5).numpy()
    np.testing.assert_allclose(got.cpu().numpy(), ref, rtol=2e-5, atol=2e-5)


def test_mfma_path_is_used_for_the_big_shapes(ops):
    assert ops.conv_order(128, 0, 128) == 1 and ops.conv_order(128, 128, 128) == 1 and ops.conv_order(16, 0, 64) == 1
    assert ops.conv_order(1, 0, 16) == 0 and ops.conv_order(128, 0, 1) == 0 and ops.conv_order(16, 0, 8) == 0


@pytest.mark.parametrize('c_in,c_out', [(16, 64), (128, 128), (1, 1), (64, 32)])
def test_stride2_conv_bit_exact(ops, scene, c_in, c_out):
    rng = np.random.default_rng(c_in + c_out)
    lvl, up, table = scene['lvl'], scene['up'], scene['k2']
    x = rng.normal(size=(lvl.n, c_in)).astype(np.float32)
    w = (rng.normal(size=(8, c_in, c_out)) / np.sqrt(4 * c_in)).astype(np.float32)
    b = rng.normal(size=c_out).astype(np.float32)
    child_row = np.ascontiguousarray(table.T)                    # [m, 8], the layout the pyramid keeps
    got = ops.conv_f32(_cuda(x), _cuda(w), c_out, up.n, nbr=_cuda(child_row), n_offsets=8, nbr_ks=1, nbr_os=8,
                       bias=_cuda(b), act=ops.ACT_RELU)
    want = sc.conv_chain(x, table, w, b, up.n, act=sc.ACT_RELU, order=ops.conv_order(c_in, 0, c_out, 8, 1, up.n))
    assert (_bits(got.cpu().numpy()) == _bits(want)).all()


@pytest.mark.parametrize('c_in,c_out', [(128, 128), (128, 32), (1, 128), (64, 16), (1, 1)])
def test_transposed_and_generative_bit_exact(ops, scene, c_in, c_out):
    rng = np.random.default_rng(7 * c_in + c_out)
    lvl, up = scene['lvl'], scene['up']
    child_row = np.ascontiguousarray(scene['k2'].T)              # [m, 8] rows of the fine level
    x = rng.normal(size=(up.n, c_in)).astype(np.float32)
    w = (rng.normal(size=(8, c_in, c_out)) / np.sqrt(c_in)).astype(np.float32)
    b = rng.normal(size=c_out).astype(np.float32)
    slope = torch.tensor([0.3], device='cuda')
    order = ops.conv_order(c_in, 0, c_out)
    # transposed onto the existing fine level
    got = ops.conv_f32(_cuda(x), _cuda(w), c_out, up.n, groups=8, out_map=_cuda(child_row), om_os=8, om_gs=1,
                       out_rows=lvl.n, bias=_cuda(b), act=ops.ACT_PRELU, slope=slope)
    want = np.zeros((lvl.n, c_out), np.float32)
    gen = np.zeros((8 * up.n, c_out), np.float32)
    for g in range(8):
        y = sc.conv_chain(x, None, w[g], b, up.n, act=sc.ACT_PRELU, slope=0.3, order=order)
        rows = child_row[:, g]
        want[rows[rows >= 0]] = y[rows >= 0]
        gen[g::8] = y
    assert (_bits(got.cpu().numpy()) == _bits(want)).all()
    # generative: all 8 children, row = 8 * parent + octant
    got = ops.conv_f32(_cuda(x), _cuda(w), c_out, up.n, groups=8, bias=_cuda(b), act=ops.ACT_PRELU, slope=slope)
    assert (_bits(got.cpu().numpy()) == _bits(gen)).all()
    # ... and it equals the oracle's transposed kernel map formulation
    g_lvl = oc.generated(up)
    ref = sc.conv_mm(torch.from_numpy(x), oc.transposed_map(up, g_lvl), torch.from_numpy(w), torch.from_numpy(b),
                     g_lvl.n, sc.ACT_PRELU, 0.3).numpy()
    np.testing.assert_allclose(got.cpu().numpy(), ref, rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize('c1,c2,c_out', [(1, 0, 64), (64, 0, 128), (128, 128, 128), (128, 0, 1), (16, 0, 8), (8, 0, 1)])
def test_pointwise_linear_bit_exact(ops, c1, c2, c_out):
    rng = np.random.default_rng(c1 + c2 + c_out)
    for n in (1, 127, 128, 129, 5000):
        x1 = rng.normal(size=(n, c1)).astype(np.float32)
        x2 = rng.normal(size=(n, c2)).astype(np.float32) if c2 else None
        w = rng.normal(size=(c1 + c2, c_out)).astype(np.float32)
        b = rng.normal(size=c_out).astype(np.float32)
        got = ops.conv_f32(_cuda(x1), _cuda(w), c_out, n, x2=None if x2 is None else _cuda(x2), bias=_cuda(b))
        want = sc.conv_chain(x1, None, w, b, n, x2=x2, order=ops.conv_order(c1, c2, c_out))
        assert (_bits(got.cpu().numpy()) == _bits(want)).all()


def test_isolated_rows_and_empty_input(ops):
    # rows without any neighbour get act(bias); zero rows is a no-op
    n, c = 300, 64
    table = np.full((27, n), -1, np.int32)
    w = np.ones((27, c, c), np.float32)
    b = np.arange(c, dtype=np.float32) - 30
    got = ops.conv_f32(_cuda(np.ones((n, c), np.float32)), _cuda(w), c, n, nbr=_cuda(table), n_offsets=27, nbr_ks=n,
                       nbr_os=1, bias=_cuda(b), act=ops.ACT_RELU)
    assert (got.cpu().numpy() == np.maximum(b, 0)[None]).all()
    out = ops.conv_f32(torch.zeros((0, c), device='cuda'), _cuda(w[:1]), c, 0)
    assert out.shape == (0, c)


def test_argument_checks(ops):
    x = torch.zeros((4, 8), device='cuda')
    w = torch.zeros((8, 8), device='cuda')
    with pytest.raises(ops.FpccError):
        ops.conv_f32(x, torch.zeros((27, 8, 8), device='cuda'), 8, 4, n_offsets=27)        # no table for 27 offsets
    with pytest.raises(ops.FpccError):
        ops.conv_f32(x, w, 8, 4, act=ops.ACT_PRELU)                                          # PReLU without slope
    with pytest.raises((ops.FpccError, TypeError)):
        ops.conv_f32(x.cpu(), w, 8, 4)                                                       # host tensor


def test_single_channel_conv3_two_phase_bit_exact(ops, scene):
    """C_out == 1: per-row dot products on the MFMA kernel + scalar gather-sum == oracle order 2"""
    rng = np.random.default_rng(77)
    lvl, table = scene['lvl'], scene['k3']
    n = lvl.n
    for c_in in (32, 128):
        x = rng.normal(size=(n, c_in)).astype(np.float32)
        w = (rng.normal(size=(27, c_in, 1)) / np.sqrt(13 * c_in)).astype(np.float32)
        b = rng.normal(size=1).astype(np.float32)
        wt = np.zeros((c_in, 32), np.float32)
        wt[:, :27] = w[:, :, 0].T
        y = ops.conv_f32(_cuda(x), _cuda(wt), 32, n)
        got = ops.gather_sum(y, _cuda(table), 27, n, 1, n, bias=_cuda(b), clip=0.7)
        want = sc.conv_chain(x, table, w, b, n, clip=0.7, order=2)
        assert (_bits(got.cpu().numpy()) == _bits(want)).all()


def test_engine_layers_match_oracle_orders(ops, scene):
    """the fused evaluations chosen inside fastpcc_amd.engine (packed generative GEMM, two-phase conv3 -> 1) agree bit for
    bit with the oracle when it is told the order engine.summation_order reports"""
    from fastpcc_amd import engine as ME
    rng = np.random.default_rng(5)
    lvl = scene['lvl']
    coords = torch.from_numpy(lvl.coords).to(torch.int32).cuda()
    coords2 = coords.clone()
    coords2[:, 1:] *= 2                      # the same cloud at tensor stride 2, so that it can be upsampled
    with torch.no_grad():
        for c_in, c_out in ((64, 16), (128, 32), (16, 4), (1, 1)):
            cm = ME.CoordinateManager()
            x = rng.normal(size=(lvl.n, c_in)).astype(np.float32)
            st = ME.SparseTensor(_cuda(x), coordinates=coords2, tensor_stride=2, coordinate_manager=cm)
            conv = ME.MinkowskiGenerativeConvolutionTranspose(c_in, c_out, kernel_size=2, stride=2, bias=True).cuda()
            out = conv(st)
            w = conv.kernel.detach().cpu().numpy()
            b = conv.bias.detach().cpu().numpy().reshape(-1)
            order = ME.summation_order('gen', c_in, 0, c_out, lvl.n)
            want = np.zeros((8 * lvl.n, c_out), np.float32)
            for g in range(8):
                want[g::8] = sc.conv_chain(x, None, w[g], b, lvl.n, order=order)
            assert (_bits(out.F.cpu().numpy()) == _bits(want)).all(), (c_in, c_out)
        for c_in in (32, 128, 8):
            cm = ME.CoordinateManager()
            x = rng.normal(size=(lvl.n, c_in)).astype(np.float32)
            st = ME.SparseTensor(_cuda(x), coordinates=coords, coordinate_manager=cm)
            conv = ME.MinkowskiConvolution(c_in, 1, kernel_size=3, stride=1, bias=True).cuda()
            out = conv(st)
            want = sc.conv_chain(x, scene['k3'], conv.kernel.detach().cpu().numpy(),
                                 conv.bias.detach().cpu().numpy().reshape(-1), lvl.n,
                                 order=ME.summation_order('k3', c_in, 0, 1, lvl.n))
            assert (_bits(out.F.cpu().numpy()) == _bits(want)).all(), c_in


@pytest.mark.parametrize('c1,c2,c_out', [(128, 0, 128), (128, 128, 128), (64, 0, 64), (16, 0, 64), (48, 0, 32), (48, 0, 64)])
def test_row_order_changes_no_result(ops, c1, c2, c_out):
    """neighbour-pattern row order (fpcc_conv_row_keys + sort): a permutation, windows respected, and the convolution in
    that order is bit-identical to the natural order and to the oracle chain"""
    xyz = surface_cloud(33, 256, 400000)
    lvl = oc.Level(batched(xyz), 1)
    n = lvl.n - 37                                   # ragged tail tile
    table = oc.dense_table(oc.kernel_map(lvl, lvl, 3), lvl.n)[:, :n].copy()
    nbr = _cuda(table)
    windowed = ops.conv_row_order(nbr, 27, n, 1, n, 11, heaviest_first=False).cpu().numpy()
    assert sorted(windowed.tolist()) == list(range(n))
    assert ((windowed >> 11) == (np.arange(n) >> 11)).all()             # rows stay inside their window of 2048
    # blocks of 32 rows in the new order need fewer (block, offset) products
    pres = table >= 0
    def per_block(perm, rows=32):
        pad = (-len(perm)) % rows
        p = np.concatenate((pres[:, perm], np.zeros((27, pad), bool)), 1).reshape(27, -1, rows)
        return p.any(2).sum(0)
    assert per_block(windowed).sum() < 0.85 * per_block(np.arange(n)).sum()
    # ... and, tile by tile (64 rows from 32 Ki rows up), the tiles that execute most offsets come first
    order = ops.conv_row_order(nbr, 27, n, 1, n, 11)
    o = order.cpu().numpy()
    assert sorted(o.tolist()) == list(range(n))
    group = 64 if n >= 32 * 1024 else 32
    full = n // group * group
    weights = per_block(o[:full], group)
    assert (np.diff(weights) <= 0).all() and weights[0] > weights[-1]
    assert sorted(map(tuple, o[:full].reshape(-1, group).tolist())) == sorted(map(tuple, windowed[:full].reshape(-1, group).tolist()))
    assert (o[full:] == windowed[full:]).all()                         # the ragged tail stays last
    rng = np.random.default_rng(c1 + c_out)
    x1 = rng.normal(size=(lvl.n, c1)).astype(np.float32)
    x2 = rng.normal(size=(lvl.n, c2)).astype(np.float32) if c2 else None
    w = (rng.normal(size=(27, c1 + c2, c_out)) / np.sqrt(13 * (c1 + c2))).astype(np.float32)
    b = rng.normal(size=c_out).astype(np.float32)
    slope = torch.tensor([0.1], device='cuda')
    args = dict(x2=None if x2 is None else _cuda(x2), nbr=nbr, n_offsets=27, nbr_ks=n, nbr_os=1, bias=_cuda(b),
                act=ops.ACT_PRELU, slope=slope)
    plain = ops.conv_f32(_cuda(x1), _cuda(w), c_out, n, **args)
    got = ops.conv_f32(_cuda(x1), _cuda(w), c_out, n, row_order=order, **args)
    assert torch.equal(got, plain)
    order_of_shape = ops.conv_order(c1, c2, c_out, 27, 1, n)
    assert order_of_shape in (1, 3)
    want = sc.conv_chain(x1, table, w, b, n, x2=x2, act=sc.ACT_PRELU, slope=0.1, order=order_of_shape)
    assert (_bits(got.cpu().numpy()) == _bits(want)).all()


def test_row_order_on_a_small_grouped_map(ops):
    xyz = surface_cloud(35, 64, 9000)
    lvl = oc.Level(batched(xyz), 1)
    n = min(lvl.n, 5000)
    table = oc.dense_table(oc.kernel_map(lvl, lvl, 3), lvl.n)[:, :n].copy()
    nbr = _cuda(table)
    order = ops.conv_row_order(nbr, 27, n, 1, n, 8)
    rng = np.random.default_rng(3)
    x = _cuda(rng.normal(size=(lvl.n, 128)).astype(np.float32))
    w = _cuda((rng.normal(size=(27, 128, 128)) / 40).astype(np.float32))
    a = ops.conv_f32(x, w, 128, n, nbr=nbr, n_offsets=27, nbr_ks=n, nbr_os=1)
    b = ops.conv_f32(x, w, 128, n, nbr=nbr, n_offsets=27, nbr_ks=n, nbr_os=1, row_order=order)
    assert torch.equal(a, b)


@pytest.mark.parametrize('c1,c2,c_out,n_off', [(128, 0, 128, 27), (128, 128, 128, 27), (64, 0, 64, 27), (128, 0, 32, 27),
                                               (128, 0, 128, 8), (32, 0, 64, 8)])
@pytest.mark.parametrize('rows', [1, 37, 260, 4470])
def test_small_maps_are_grouped_like_large_ones(ops, c1, c2, c_out, n_off, rows):
    """the summation order of a multi-offset MFMA shape is order 3 at every row count (numerics version 2: no row thresholds);
    without packed weights the caller's workspace receives the packed copy"""
    xyz = surface_cloud(35, 64, 9000)
    lvl = oc.Level(batched(xyz), 1)
    n = min(rows, lvl.n)
    if n_off == 27:
        table = oc.dense_table(oc.kernel_map(lvl, lvl, 3), lvl.n)[:, :n].copy()
        n_in = lvl.n
    else:
        up = oc.strided(lvl)
        n = min(rows, up.n)
        table = oc.dense_table(oc.kernel_map(lvl, up, 2), up.n)[:, :n].copy()
        n_in = lvl.n
    assert ops.conv_order(c1, c2, c_out, n_off, 1, n) == 3 == ops.conv_order(c1, c2, c_out, n_off, 1, 10 ** 7)
    assert ops.lib().fpcc_conv_f32_ws_bytes(c1, c2, c_out, n_off, 1, n) == n_off * (c1 + c2) * c_out * 4
    rng = np.random.default_rng(c1 + c_out + rows)
    x1 = rng.normal(size=(n_in, c1)).astype(np.float32)
    x2 = rng.normal(size=(n_in, c2)).astype(np.float32) if c2 else None
    w = (rng.normal(size=(n_off, c1 + c2, c_out)) / np.sqrt(n_off / 2 * (c1 + c2))).astype(np.float32)
    b = rng.normal(size=c_out).astype(np.float32)
    slope = torch.tensor([0.3], device='cuda')
    args = dict(x2=None if x2 is None else _cuda(x2), nbr=_cuda(table), n_offsets=n_off, nbr_ks=n, nbr_os=1, bias=_cuda(b),
                act=ops.ACT_PRELU, slope=slope, clip=2.0)
    got = ops.conv_f32(_cuda(x1), _cuda(w), c_out, n, **args)
    want = sc.conv_chain(x1, table, w, b, n, x2=x2, act=sc.ACT_PRELU, slope=0.3, clip=2.0, order=3)
    assert (_bits(got.cpu().numpy()) == _bits(want)).all()
    assert torch.equal(got, ops.conv_f32(_cuda(x1), _cuda(w), c_out, n, pack=True, **args))


def test_grouped_shape_without_packed_weights_needs_a_workspace(ops):
    L = ops.lib()
    x = torch.zeros((64, 128), device='cuda')
    w = torch.zeros((27, 128, 128), device='cuda')
    nbr = torch.full((27, 64), -1, dtype=torch.int32, device='cuda')
    out = torch.empty((64, 128), device='cuda')
    rc = L.fpcc_conv_f32(x.data_ptr(), 128, 128, None, 0, 0, nbr.data_ptr(), 27, 64, 1, w.data_ptr(), None, 128, 1, None, 1, 1,
                         out.data_ptr(), 128, 64, 0, None, 0.0, None, None, 0, None)
    assert rc != 0 and b'workspace' in L.fpcc_last_error()


@pytest.mark.parametrize('n', [1, 31, 64, 65, 200])
def test_row_order_on_tiny_maps(ops, n):
    """fewer rows than one or two tile groups: the heaviest-first regrouping degenerates gracefully"""
    xyz = surface_cloud(36, 32, 2000)
    lvl = oc.Level(batched(xyz), 1)
    table = oc.dense_table(oc.kernel_map(lvl, lvl, 3), lvl.n)[:, :n].copy()
    table[table >= n] = -1
    order = ops.conv_row_order(_cuda(table), 27, n, 1, n, 5)
    assert sorted(order.cpu().tolist()) == list(range(n))


# ---- wave-autonomous kernel (packed weights): every tuning must give the workgroup-tiled kernel's bits -------------
@pytest.fixture()
def knobs(ops):
    saved = [ops.conv_set_tuning(k, v) for k, v in ((ops.KNOB_WAVE_ON, 1), (ops.KNOB_WAVE_NBW, 0), (ops.KNOB_WAVE_SB, 1))]
    yield ops
    for k, v in zip((ops.KNOB_WAVE_ON, ops.KNOB_WAVE_NBW, ops.KNOB_WAVE_SB), saved):
        ops.conv_set_tuning(k, v)


@pytest.mark.parametrize('c1,c2,c_out', [(128, 0, 128), (128, 128, 128), (64, 0, 128), (128, 0, 64), (64, 0, 64), (32, 0, 32),
                                         (96, 0, 32), (32, 32, 64)])
def test_wave_kernel_conv3_equals_tiled_kernel_and_oracle(knobs, scene, request, c1, c2, c_out):
    ops = knobs
    rng = np.random.default_rng(c1 * 7 + c2 * 3 + c_out)
    lvl, table = scene['lvl'], scene['k3']
    n = lvl.n
    x1 = _cuda(rng.normal(size=(n, c1)).astype(np.float32))
    x2 = _cuda(rng.normal(size=(n, c2)).astype(np.float32)) if c2 else None
    w = _cuda((rng.normal(size=(27, c1 + c2, c_out)) / np.sqrt(13 * (c1 + c2))).astype(np.float32))
    b = _cuda(rng.normal(size=c_out).astype(np.float32))
    slope = torch.tensor([0.2], device='cuda')
    nbr = _cuda(table)
    order = ops.conv_row_order(nbr, 27, n, 1, n, 13)
    kw = dict(x2=x2, nbr=nbr, n_offsets=27, nbr_ks=n, nbr_os=1, bias=b, act=ops.ACT_PRELU, slope=slope, clip=1.5)
    # the production order of these shapes is 3 (grouped); the order-1 kernels stay reachable for A/B experiments (knob 7 under
    # FPCC_EXPERIMENT=1) and must keep agreeing with each other and with the oracle's order-1 chain
    import os
    os.environ['FPCC_EXPERIMENT'] = '1'
    if (c1 + c2) % 32 == 0 and c1 % 32 == 0:
        for gnbw in (1, 2, 0):
            ops.conv_set_tuning(ops.KNOB_GROUPED_NBW, gnbw)
            for ro in (None, order):
                got3 = ops.conv_f32(x1, w, c_out, n, row_order=ro, pack=True, **kw).cpu().numpy()
                want3 = sc.conv_chain(x1.cpu().numpy(), table, w.cpu().numpy(), b.cpu().numpy(), n, x2=None if x2 is None else x2.cpu().numpy(),
                                      act=sc.ACT_PRELU, slope=0.2, clip=1.5, order=3)
                assert (_bits(got3) == _bits(want3)).all(), ('grouped', gnbw, ro is not None)
    ops.conv_set_tuning(ops.KNOB_GROUPED_OFF, 1)
    request.addfinalizer(lambda: ops.conv_set_tuning(ops.KNOB_GROUPED_OFF, 0))
    base = ops.conv_f32(x1, w, c_out, n, **kw).cpu().numpy()
    want = sc.conv_chain(x1.cpu().numpy(), table, w.cpu().numpy(), b.cpu().numpy(), n, x2=None if x2 is None else x2.cpu().numpy(),
                         act=sc.ACT_PRELU, slope=0.2, clip=1.5, order=1)
    assert (_bits(base) == _bits(want)).all()
    assert ops.packed_weights(w, c1, c2, c_out, 27, 1) is not None
    for nbw, sb in ((1, 0), (1, 1), (2, 0), (2, 1), (4, 0), (4, 1), (0, 1)):
        ops.conv_set_tuning(ops.KNOB_WAVE_NBW, nbw)
        ops.conv_set_tuning(ops.KNOB_WAVE_SB, sb)
        for ro in (None, order):
            got = ops.conv_f32(x1, w, c_out, n, row_order=ro, pack=True, **kw).cpu().numpy()
            assert (_bits(got) == _bits(base)).all(), (nbw, sb, ro is not None)
    if c_out >= 64:
        # 64 x 64 wave tiles (two row blocks x two column blocks per wave), forced on for this map size
        ops.conv_set_tuning(ops.KNOB_WAVE_NBW, 0)
        before = ops.conv_set_tuning(ops.KNOB_WAVE22_ROWS, 1)
        try:
            for sb in (0, 1):
                ops.conv_set_tuning(ops.KNOB_WAVE_SB, sb)
                for ro in (None, order):
                    got = ops.conv_f32(x1, w, c_out, n, row_order=ro, pack=True, **kw).cpu().numpy()
                    assert (_bits(got) == _bits(base)).all(), ('2x2', sb, ro is not None)
        finally:
            ops.conv_set_tuning(ops.KNOB_WAVE22_ROWS, before)


@pytest.mark.parametrize('n', [1, 31, 32, 33, 127, 128, 129, 4999])
def test_wave_kernel_ragged_row_counts_pointwise(knobs, n):
    ops = knobs
    rng = np.random.default_rng(n)
    for c1, c2, c_out in ((128, 0, 128), (64, 64, 32), (32, 0, 64)):
        x1 = _cuda(rng.normal(size=(n, c1)).astype(np.float32))
        x2 = _cuda(rng.normal(size=(n, c2)).astype(np.float32)) if c2 else None
        w = _cuda(rng.normal(size=(c1 + c2, c_out)).astype(np.float32))
        b = _cuda(rng.normal(size=c_out).astype(np.float32))
        base = ops.conv_f32(x1, w, c_out, n, x2=x2, bias=b).cpu().numpy()
        for nbw in (1, 2, 4, 0):
            ops.conv_set_tuning(ops.KNOB_WAVE_NBW, nbw)
            got = ops.conv_f32(x1, w, c_out, n, x2=x2, bias=b, pack=True).cpu().numpy()
            assert (_bits(got) == _bits(base)).all(), (c1, c2, c_out, nbw)


def test_wave_kernel_strided_transposed_generative(knobs, scene):
    ops = knobs
    rng = np.random.default_rng(99)
    lvl, up = scene['lvl'], scene['up']
    child_row = _cuda(np.ascontiguousarray(scene['k2'].T))
    slope = torch.tensor([0.3], device='cuda')
    for c_in, c_out in ((64, 64), (128, 128), (32, 32), (128, 32)):
        xf = _cuda(rng.normal(size=(lvl.n, c_in)).astype(np.float32))
        xc = _cuda(rng.normal(size=(up.n, c_in)).astype(np.float32))
        w = _cuda((rng.normal(size=(8, c_in, c_out)) / np.sqrt(c_in)).astype(np.float32))
        b = _cuda(rng.normal(size=c_out).astype(np.float32))
        cases = [
            (xf, dict(n_out=up.n, nbr=child_row, n_offsets=8, nbr_ks=1, nbr_os=8, bias=b, act=ops.ACT_RELU)),        # stride 2
            (xc, dict(n_out=up.n, groups=8, out_map=child_row, om_os=8, om_gs=1, out_rows=lvl.n, bias=b,
                      act=ops.ACT_PRELU, slope=slope)),                                                               # transposed
            (xc, dict(n_out=up.n, groups=8, bias=b, act=ops.ACT_PRELU, slope=slope)),                                 # generative
        ]
        for x, kw in cases:
            n_out = kw.pop('n_out')
            rows = kw.get('out_rows')
            def run(**extra):
                out = torch.zeros((rows, c_out), device='cuda') if rows else None      # rows without a parent stay untouched
                return ops.conv_f32(x, w, c_out, n_out, out=out, **{k: v for k, v in kw.items() if k != 'out_rows'}, **extra).cpu().numpy()
            base = run()
            for nbw in (1, 2, 4, 0):
                ops.conv_set_tuning(ops.KNOB_WAVE_NBW, nbw)
                assert (_bits(run(pack=True)) == _bits(base)).all(), (c_in, c_out, nbw, sorted(kw))
            before = ops.conv_set_tuning(ops.KNOB_WAVE22_ROWS, 1)         # 64 x 64 wave tiles where the shape has them
            try:
                assert (_bits(run(pack=True)) == _bits(base)).all(), (c_in, c_out, '2x2', sorted(kw))
            finally:
                ops.conv_set_tuning(ops.KNOB_WAVE22_ROWS, before)


def test_packed_weights_follow_in_place_updates(knobs):
    ops = knobs
    x = torch.randn((300, 64), device='cuda')
    w = torch.randn((64, 64), device='cuda')
    a = ops.conv_f32(x, w, 64, 300, pack=True)
    w.mul_(2.0)                                     # same storage, new version: the packed copy must be rebuilt
    b = ops.conv_f32(x, w, 64, 300, pack=True)
    assert torch.equal(b, ops.conv_f32(x, w, 64, 300)) and not torch.equal(a, b)


@pytest.mark.parametrize('n', [1, 33, 4999, 70001])
@pytest.mark.parametrize('c1,c2,c_out', [(128, 0, 128), (128, 128, 128), (64, 0, 128), (128, 0, 64), (64, 0, 64), (32, 0, 32), (32, 0, 128),
                                         (64, 64, 32), (256, 0, 128), (128, 0, 32), (64, 0, 96)])
def test_persistent_pointwise_kernel_equals_the_other_kernels(knobs, n, c1, c2, c_out):
    """per-point layers on large maps run on k_pointwise_wave (weights in registers, many row blocks per wave): forced on for
    every row count here, it must give the bits of the workgroup-tiled kernel, with bias / PReLU / clamp and two sources"""
    ops = knobs
    rng = np.random.default_rng(n + c1 + 3 * c2 + 5 * c_out)
    x1 = _cuda(rng.normal(size=(n, c1)).astype(np.float32))
    x2 = _cuda(rng.normal(size=(n, c2)).astype(np.float32)) if c2 else None
    w = _cuda((rng.normal(size=(c1 + c2, c_out)) / np.sqrt(c1 + c2)).astype(np.float32))
    b = _cuda(rng.normal(size=c_out).astype(np.float32))
    slope = torch.tensor([0.15], device='cuda')
    kw = dict(x2=x2, bias=b, act=ops.ACT_PRELU, slope=slope, clip=2.0)
    base = ops.conv_f32(x1, w, c_out, n, **kw).cpu().numpy()                       # workgroup-tiled kernel (unpacked weights)
    before = ops.conv_set_tuning(ops.KNOB_POINTWISE_ROWS, 1)
    try:
        got = ops.conv_f32(x1, w, c_out, n, pack=True, **kw).cpu().numpy()
        ops.conv_set_tuning(ops.KNOB_POINTWISE_ROWS, 0)
        wave = ops.conv_f32(x1, w, c_out, n, pack=True, **kw).cpu().numpy()
    finally:
        ops.conv_set_tuning(ops.KNOB_POINTWISE_ROWS, before)
    assert (_bits(wave) == _bits(base)).all()
    assert (_bits(got) == _bits(base)).all()


@pytest.mark.parametrize('n', [1, 2, 63, 1023, 1024, 1025, 4999, 16384])
def test_group_counting_order_equals_a_sort(ops, n):
    """fpcc_conv_group_order: the stable one-launch counting sort of the heaviest-first tile order gives the permutation of a
    64-bit sort of the same keys ((lacking offsets) << 32 | group)"""
    g = torch.Generator().manual_seed(n)
    lack = torch.randint(0, 28, (n,), generator=g)
    if n > 10:
        lack[: n // 3] = 5                                             # long runs of equal weights: stability matters
    keys = ((lack.to(torch.int64) << 32) | torch.arange(n, dtype=torch.int64)).cuda()
    perm = torch.empty(n, dtype=torch.int32, device='cuda')
    from fastpcc_amd.hipops import lib, _ok, _stream
    _ok(lib().fpcc_conv_group_order(keys.data_ptr(), n, perm.data_ptr(), _stream()))
    want = torch.argsort(keys)
    assert torch.equal(perm.long(), want)


@pytest.mark.parametrize('c_out', [16, 32, 8])
def test_constant_one_first_layer_from_presence_masks(ops, scene, c_out):
    """fpcc_conv_ones_k3_f32 + fpcc_mask27_from_parent: the codec's first layer (every voxel carries the feature 1) evaluated
    from 27-bit neighbour masks equals the general kernel on the neighbour table bit for bit; the masks equal the table's"""
    rng = np.random.default_rng(c_out)
    lvl, table = scene['lvl'], scene['k3']
    n = lvl.n
    nbr = _cuda(table)
    masks = torch.zeros(n, dtype=torch.int32, device='cuda')
    for k in range(27):
        masks |= (nbr[k] >= 0).to(torch.int32) << k
    w = _cuda((rng.normal(size=(27, 1, c_out)) / 4).astype(np.float32))
    b = _cuda(rng.normal(size=c_out).astype(np.float32))
    slope = torch.tensor([0.3], device='cuda')
    ones = torch.ones((n, 1), device='cuda')
    want = ops.conv_f32(ones, w, c_out, n, nbr=nbr, n_offsets=27, nbr_ks=n, nbr_os=1, bias=b, act=ops.ACT_PRELU, slope=slope)
    got = ops.conv_ones_k3(masks, w, c_out, bias=b, act=ops.ACT_PRELU, slope=slope)
    assert (_bits(got.cpu().numpy()) == _bits(want.cpu().numpy())).all()
    # masks from the parent level == masks of the table
    from fastpcc_amd import engine as ME
    cm = ME.CoordinateManager(D=3)
    key, _ = cm.insert_and_map(torch.from_numpy(lvl.coords.astype(np.int32)).cuda(), 1)
    m = cm._map(key)
    derived = cm._mask27(m)
    assert derived is not None and torch.equal(derived, masks)


@pytest.mark.parametrize('c_out,n', [(16, 1), (16, 100003), (4, 777), (12, 5000), (24, 33333), (32, 70001), (8, 4099)])
def test_constant_one_first_layer_on_random_masks(ops, c_out, n):
    """the register-weight form of fpcc_conv_ones_k3_f32 (a thread owns 8 or 4 columns, absent neighbours enter as fmaf(0, w, acc)):
    the ascending-offset fp32 chain over the present neighbours, bit for bit, for every column split, ragged sizes, empty masks and
    weights of either sign including -0"""
    rng = np.random.default_rng(c_out * 1000 + n)
    masks = rng.integers(0, 1 << 27, n, dtype=np.int64)
    masks[rng.random(n) < 0.05] = 0
    masks[rng.random(n) < 0.05] = (1 << 27) - 1
    w = (rng.normal(size=(27, 1, c_out)) / 4).astype(np.float32)
    w[rng.random(w.shape) < 0.05] = -0.0
    b = rng.normal(size=c_out).astype(np.float32)
    acc = np.zeros((n, c_out), np.float32)
    for k in range(27):
        bit = ((masks >> k) & 1).astype(bool)[:, None]
        acc = np.where(bit, acc + w[k, 0][None, :], acc).astype(np.float32)
    want = (acc + b[None, :]).astype(np.float32)
    want = np.where(want < 0, (want * np.float32(0.3)).astype(np.float32), want)
    slope = torch.tensor([0.3], device='cuda')
    got = ops.conv_ones_k3(torch.from_numpy(masks.astype(np.int32)).cuda(), _cuda(w), c_out, bias=_cuda(b), act=ops.ACT_PRELU, slope=slope)
    assert (_bits(got.cpu().numpy()) == _bits(want)).all()


@pytest.mark.parametrize('c1,c2,c_out,n_off', [(128, 0, 128, 27), (128, 128, 128, 27), (64, 0, 64, 27), (64, 0, 128, 8), (32, 0, 32, 27)])
@pytest.mark.parametrize('nbw', [1, 2])
def test_grouped_evaluation_is_order_3(ops, scene, c1, c2, c_out, n_off, nbw):
    """summation order 3 (four fixed offset groups, one wave each, partial sums added in group order): bit for bit the oracle's
    order-3 chain, with packed weights and with the pack-into-workspace path, in natural and in pattern row order"""
    rng = np.random.default_rng(c1 + c2 + c_out + n_off)
    lvl = scene['lvl']
    table = scene['k3'] if n_off == 27 else scene['k2']
    n = table.shape[1]
    n_in = lvl.n
    x1 = rng.normal(size=(n_in, c1)).astype(np.float32)
    x2 = rng.normal(size=(n_in, c2)).astype(np.float32) if c2 else None
    w = (rng.normal(size=(n_off, c1 + c2, c_out)) / np.sqrt(n_off / 2 * (c1 + c2))).astype(np.float32)
    b = rng.normal(size=c_out).astype(np.float32)
    slope = torch.tensor([0.25], device='cuda')
    saved = [ops.conv_set_tuning(k, v) for k, v in ((ops.KNOB_GROUPED_NBW, nbw),)]
    try:
        assert ops.conv_order(c1, c2, c_out, n_off, 1, n) == 3
        kw = dict(x2=None if x2 is None else _cuda(x2), nbr=_cuda(table), n_offsets=n_off, nbr_ks=n, nbr_os=1, bias=_cuda(b),
                  act=ops.ACT_PRELU, slope=slope, clip=1.9)
        packed = ops.conv_f32(_cuda(x1), _cuda(w), c_out, n, pack=True, **kw)
        plain = ops.conv_f32(_cuda(x1), _cuda(w), c_out, n, **kw)
        order = ops.conv_row_order(kw['nbr'], n_off, n, 1, n) if n_off == 27 else None
        ordered = ops.conv_f32(_cuda(x1), _cuda(w), c_out, n, pack=True, row_order=order, **kw)
    finally:
        for k, v in zip((ops.KNOB_GROUPED_NBW,), saved):
            ops.conv_set_tuning(k, v)
    want = sc.conv_chain(x1, table, w, b, n, x2=x2, act=sc.ACT_PRELU, slope=0.25, clip=1.9, order=3)
    assert (_bits(packed.cpu().numpy()) == _bits(want)).all()
    assert (_bits(plain.cpu().numpy()) == _bits(want)).all()
    assert (_bits(ordered.cpu().numpy()) == _bits(want)).all()


@pytest.mark.parametrize('c1,c2,c_out,n_off', [(128, 0, 128, 27), (128, 128, 128, 27), (64, 0, 64, 27), (64, 0, 128, 8), (32, 0, 32, 27), (96, 0, 32, 27)])
@pytest.mark.parametrize('sparse', [False, True])
def test_folded_evaluation_is_order_3_too(ops, scene, c1, c2, c_out, n_off, sparse):
    """the folded form of the grouped evaluation (ONE wave per unit adds the four offset groups up itself; the form large maps take) gives
    the oracle's order-3 bits and the four-wave form's -- also when whole offset groups are absent from a block (sparse: the table keeps
    the offsets of one or two groups per run of 48 rows, so leading, middle and trailing groups go missing) and with a row order"""
    rng = np.random.default_rng(c1 + c2 + c_out + n_off + 1000 * sparse)
    lvl = scene['lvl']
    table = (scene['k3'] if n_off == 27 else scene['k2']).copy()
    n = table.shape[1]
    if sparse:
        begins = [(g * n_off + 3) // 4 for g in range(5)]
        for r0 in range(0, n, 48):
            keep = [(r0 // 48) % 4] if (r0 // 96) % 2 else [(r0 // 48) % 4, (r0 // 48 + 2) % 4]
            for g in range(4):
                if g not in keep:
                    table[begins[g]:begins[g + 1], r0:r0 + 48] = -1
        table[:, 7] = -1                                                         # and a row without any neighbour
    n_in = lvl.n
    x1 = rng.normal(size=(n_in, c1)).astype(np.float32)
    x2 = rng.normal(size=(n_in, c2)).astype(np.float32) if c2 else None
    w = (rng.normal(size=(n_off, c1 + c2, c_out)) / np.sqrt(n_off / 2 * (c1 + c2))).astype(np.float32)
    b = rng.normal(size=c_out).astype(np.float32)
    slope = torch.tensor([0.25], device='cuda')
    kw = dict(x2=None if x2 is None else _cuda(x2), nbr=_cuda(table), n_offsets=n_off, nbr_ks=n, nbr_os=1, bias=_cuda(b),
              act=ops.ACT_PRELU, slope=slope, clip=1.9)
    order = ops.conv_row_order(kw['nbr'], n_off, n, 1, n) if n_off == 27 else None
    four_waves = ops.conv_f32(_cuda(x1), _cuda(w), c_out, n, pack=True, **kw)
    saved = ops.conv_set_tuning(ops.KNOB_GROUPED_FOLD_ROWS, 1)
    try:
        folded = ops.conv_f32(_cuda(x1), _cuda(w), c_out, n, pack=True, **kw)
        ordered = ops.conv_f32(_cuda(x1), _cuda(w), c_out, n, pack=True, row_order=order, **kw)
        plain = ops.conv_f32(_cuda(x1), _cuda(w), c_out, n, **kw)
    finally:
        ops.conv_set_tuning(ops.KNOB_GROUPED_FOLD_ROWS, saved)
    want = sc.conv_chain(x1, table, w, b, n, x2=x2, act=sc.ACT_PRELU, slope=0.25, clip=1.9, order=3)
    for got in (four_waves, folded, ordered, plain):
        assert (_bits(got.cpu().numpy()) == _bits(want)).all()


@pytest.mark.parametrize('c1,c2,c_out,n_off', [(128, 0, 128, 27), (128, 128, 128, 27), (64, 0, 64, 27), (64, 0, 128, 8), (128, 0, 64, 27), (96, 0, 64, 27)])
@pytest.mark.parametrize('sparse', [False, True])
@pytest.mark.parametrize('row_blocks', [2, 3, 4])
def test_lds_operand_kernel_is_order_3_too(ops, scene, c1, c2, c_out, n_off, sparse, row_blocks):
    """k_conv_lds (both MFMA operands staged through LDS by LDS-DMA, 2 | 3 | 4 row blocks of a workgroup in lockstep over the union of
    their offsets) leaves the oracle's order-3 bits: dense and sparse tables (whole offset groups missing from a block, blocks of one
    tile with different offset sets, a row without neighbours), natural and pattern row order, a ragged last tile"""
    rng = np.random.default_rng(c1 + c2 + c_out + n_off + 1000 * sparse + row_blocks)
    lvl = scene['lvl']
    table = (scene['k3'] if n_off == 27 else scene['k2']).copy()
    n = table.shape[1]
    if sparse:
        begins = [(g * n_off + 3) // 4 for g in range(5)]
        for r0 in range(0, n, 48):
            keep = [(r0 // 48) % 4] if (r0 // 96) % 2 else [(r0 // 48) % 4, (r0 // 48 + 2) % 4]
            for g in range(4):
                if g not in keep:
                    table[begins[g]:begins[g + 1], r0:r0 + 48] = -1
        table[:, 7] = -1
        table[:, 4096:4096 + 32 * row_blocks] = -1                                # a whole tile without any neighbour
    n_in = lvl.n
    x1 = rng.normal(size=(n_in, c1)).astype(np.float32)
    x2 = rng.normal(size=(n_in, c2)).astype(np.float32) if c2 else None
    w = (rng.normal(size=(n_off, c1 + c2, c_out)) / np.sqrt(n_off / 2 * (c1 + c2))).astype(np.float32)
    b = rng.normal(size=c_out).astype(np.float32)
    slope = torch.tensor([0.25], device='cuda')
    kw = dict(x2=None if x2 is None else _cuda(x2), nbr=_cuda(table), n_offsets=n_off, nbr_ks=n, nbr_os=1, bias=_cuda(b),
              act=ops.ACT_PRELU, slope=slope, clip=1.9)
    order = ops.conv_row_order(kw['nbr'], n_off, n, 1, n) if n_off == 27 else None
    saved = [ops.conv_set_tuning(ops.KNOB_LDS_ROWS, 1), ops.conv_set_tuning(ops.KNOB_LDS_ROW_BLOCKS, row_blocks)]
    try:
        natural = ops.conv_f32(_cuda(x1), _cuda(w), c_out, n, pack=True, **kw)
        ordered = ops.conv_f32(_cuda(x1), _cuda(w), c_out, n, pack=True, row_order=order, **kw)
        plain = ops.conv_f32(_cuda(x1), _cuda(w), c_out, n, **kw)
    finally:
        ops.conv_set_tuning(ops.KNOB_LDS_ROWS, saved[0])
        ops.conv_set_tuning(ops.KNOB_LDS_ROW_BLOCKS, saved[1])
    want = sc.conv_chain(x1, table, w, b, n, x2=x2, act=sc.ACT_PRELU, slope=0.25, clip=1.9, order=3)
    for got in (natural, ordered, plain):
        assert (_bits(got.cpu().numpy()) == _bits(want)).all()


@pytest.mark.parametrize('c1,c2,c_out', [(128, 0, 128), (64, 0, 64), (128, 128, 128), (32, 0, 32)])
def test_row_major_neighbour_table_changes_no_result(ops, scene, c1, c2, c_out):
    """the same 3x3x3 table offset-major [27, n] and row-major [n, 32] (fpcc_transpose_table_i32; what the engine hands the MFMA kernels):
    every kernel form -- four-wave grouped, folded, LDS operands -- in natural and pattern row order gives the oracle's bits from both"""
    rng = np.random.default_rng(c1 + c2 + c_out + 5)
    lvl = scene['lvl']
    table = scene['k3'].copy()
    n = table.shape[1]
    table[:, 11] = -1
    x1 = rng.normal(size=(lvl.n, c1)).astype(np.float32)
    x2 = rng.normal(size=(lvl.n, c2)).astype(np.float32) if c2 else None
    w = (rng.normal(size=(27, c1 + c2, c_out)) / np.sqrt(13 * (c1 + c2))).astype(np.float32)
    b = rng.normal(size=c_out).astype(np.float32)
    want = sc.conv_chain(x1, table, w, b, n, x2=x2, act=sc.ACT_RELU, order=3)
    nbr = _cuda(table)
    rows = ops.transpose_table(nbr, 32)
    assert rows.shape == (n, 32) and torch.equal(rows[:, :27].t().contiguous(), nbr) and bool((rows[:, 27:] == -1).all())
    order = ops.conv_row_order(nbr, 27, n, 1, n)
    base = dict(x2=None if x2 is None else _cuda(x2), bias=_cuda(b), act=ops.ACT_RELU, pack=True)
    rows_pos = rows.index_select(0, order.long())              # beside a row order the row-major table is indexed by position
    layouts = (dict(nbr=nbr, n_offsets=27, nbr_ks=n, nbr_os=1), dict(nbr=rows, n_offsets=27, nbr_ks=1, nbr_os=32))
    forms = [((ops.KNOB_GROUPED_FOLD_ROWS, 0), (ops.KNOB_PERSIST, 0)), ((ops.KNOB_GROUPED_FOLD_ROWS, 1), (ops.KNOB_PERSIST, 0)),
             # persistent forms: 40 / 17 workgroups walk the units with a stride (ragged: the counts do not divide the units)
             ((ops.KNOB_GROUPED_FOLD_ROWS, 0), (ops.KNOB_PERSIST, 40)), ((ops.KNOB_GROUPED_FOLD_ROWS, 1), (ops.KNOB_PERSIST, 17)),
             ((ops.KNOB_GROUPED_FOLD_ROWS, 1), (ops.KNOB_PERSIST, 3))]
    if c_out >= 64:
        forms += [((ops.KNOB_LDS_ROWS, 1), (ops.KNOB_LDS_ROW_BLOCKS, 2)), ((ops.KNOB_LDS_ROWS, 1), (ops.KNOB_LDS_ROW_BLOCKS, 4))]
    for form in forms:
        saved = [(k, ops.conv_set_tuning(k, v)) for k, v in form]
        try:
            for lay in layouts:
                for ro in (None, order):
                    if ro is not None and lay['nbr_ks'] == 1:
                        lay = dict(lay, nbr=rows_pos)
                    got = ops.conv_f32(_cuda(x1), _cuda(w), c_out, n, row_order=ro, **lay, **base)
                    assert (_bits(got.cpu().numpy()) == _bits(want)).all(), (form, lay['nbr_ks'], ro is not None)
        finally:
            for k, v in saved:
                ops.conv_set_tuning(k, v)


@pytest.mark.parametrize('pack', [True, False])
@pytest.mark.parametrize('c1,c2,c_out', [(32, 32, 32), (64, 0, 32), (48, 0, 64), (16, 0, 32), (32, 0, 64), (16, 16, 16), (128, 0, 64)])
def test_position_ordered_table_on_every_mfma_kernel(ops, scene, c1, c2, c_out, pack):
    """beside a row order the row-major table holds its rows in POSITION order (include/fpcc_hip.h): every kernel that takes a row
    order -- the block-tiled one, the 64-row wave units, the grouped / folded ones -- must read it that way, not by row id
    (round 4: the colour decoder's 34 -> 16 layer ran on a kernel that did not).  Same kernel, same order: same bits as from the
    offset-major table, which the oracle tests above pin."""
    if ops.conv_order(c1, c2, c_out) == 0:
        pytest.skip('not an MFMA shape')
    rng = np.random.default_rng(c1 + 3 * c2 + c_out)
    table = scene['k3'].copy()
    n = table.shape[1]
    table[:, 7] = -1
    x1 = _cuda(rng.normal(size=(n, c1)).astype(np.float32))
    x2 = _cuda(rng.normal(size=(n, c2)).astype(np.float32)) if c2 else None
    w = _cuda((rng.normal(size=(27, c1 + c2, c_out)) / np.sqrt(13 * (c1 + c2))).astype(np.float32))
    b = _cuda(rng.normal(size=c_out).astype(np.float32))
    nbr = _cuda(table)
    order = ops.conv_row_order(nbr, 27, n, 1, n)
    rows_pos = ops.transpose_table(nbr, 32).index_select(0, order.long())
    kw = dict(x2=x2, bias=b, act=ops.ACT_RELU, pack=pack, row_order=order)
    by_row = ops.conv_f32(x1, w, c_out, n, nbr=nbr, n_offsets=27, nbr_ks=n, nbr_os=1, **kw)
    by_pos = ops.conv_f32(x1, w, c_out, n, nbr=rows_pos, n_offsets=27, nbr_ks=1, nbr_os=32, **kw)
    natural = ops.conv_f32(x1, w, c_out, n, nbr=nbr, n_offsets=27, nbr_ks=n, nbr_os=1, **dict(kw, row_order=None))
    assert torch.equal(by_row, natural)
    assert torch.equal(by_pos, by_row)


def test_launch_events_recorded_inside_the_call(ops, scene):
    """fpcc_time_next_launch (hipops.set_thread_trace): the traced launch carries two events recorded inside the C call; their distance
    is the kernel's duration -- positive, and far below the time the host spends between two launches when it sleeps in between"""
    import time
    rng = np.random.default_rng(11)
    table = scene['k3']
    n = table.shape[1]
    x = _cuda(rng.normal(size=(n, 64)).astype(np.float32))
    w = _cuda((rng.normal(size=(27, 64, 64)) / 30).astype(np.float32))
    nbr = _cuda(table)
    want = ops.conv_f32(x, w, 64, n, nbr=nbr, n_offsets=27, nbr_ks=n, nbr_os=1, pack=True)
    trace = []
    ops.reserve_trace_events(8)
    ops.set_thread_trace(trace)
    try:
        for _ in range(3):
            got = ops.conv_f32(x, w, 64, n, nbr=nbr, n_offsets=27, nbr_ks=n, nbr_os=1, pack=True)
            time.sleep(0.02)                       # host time between launches must not show up in the events
    finally:
        ops.set_thread_trace(None)
    torch.cuda.synchronize()
    assert torch.equal(got, want) and len(trace) == 3
    for ev0, ev1, info in trace:
        ms = ev0.elapsed_time(ev1)
        assert 0.0 < ms < 5.0 and info['n_out'] == n and info['n_offsets'] == 27
    untraced = ops.conv_f32(x, w, 64, n, nbr=nbr, n_offsets=27, nbr_ks=n, nbr_os=1, pack=True)      # the bracket does not linger
    assert torch.equal(untraced, want) and len(trace) == 3
