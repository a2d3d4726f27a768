"""Integer kernels of libfpcc_hip.so (through the C ABI) against the oracle: exact equality everywhere."""
import numpy as np
import pytest
import torch

from fastpcc_amd.synthetic import batched, lidar_cloud
from oracle import codec_int as oi

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ops():
    from fastpcc_amd import hipops
    return hipops


@pytest.fixture(scope='module')
def cloud():
    xyz = lidar_cloud(3, beams=32, azimuths=1024)
    c = batched(xyz).astype(np.int64)
    order = np.argsort(oi.morton_encode(c[:, 1:], 'xyz', inverse=True), kind='stable')
    return c[order]


def _cuda(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    return (t if dtype is None else t.to(dtype)).cuda()


def _table_gpu(ops, in_c, out_c, ks, st):
    cap = 2 * len(in_c)
    keys = torch.zeros(cap, dtype=torch.int64, device='cuda')
    vals = torch.zeros(cap, dtype=torch.int32, device='cuda')
    ops.hash_insert_coords(keys, vals, _cuda(in_c[:, [1, 2, 3, 0]], torch.int32))
    t = ops.hash_lookup_coords(keys, vals, _cuda(out_c[:, [1, 2, 3, 0]], torch.int32), ks, st)
    return keys, vals, t


@pytest.mark.parametrize('ks,st', [((3, 3, 3), (1, 1, 1)), ((2, 2, 2), (2, 2, 2)), ((4, 4, 4), (4, 4, 4)), ((1, 1, 1), (1, 1, 1))])
def test_hash_lookup_matches_oracle(ops, cloud, ks, st):
    out_c = cloud.copy()
    if st != (1, 1, 1):
        sh = st[0].bit_length() - 1
        out_c[:, 1:] >>= sh
        out_c = np.unique(out_c, axis=0)
    keys, vals, t = _table_gpu(ops, cloud, out_c, ks, st)
    t = t.cpu().numpy()
    assert t.shape[0] % 128 == 0 and (t[len(out_c):] == 0).all()
    want = oi.kernel_table(cloud, out_c, ks, st).T + 1                     # row + 1, 0 = absent
    assert (t[:len(out_c)] == want).all()
    # a cached table is reusable, and every key is stored once
    t2 = ops.hash_lookup_coords(keys, vals, _cuda(out_c[:, [1, 2, 3, 0]], torch.int32), ks, st).cpu().numpy()
    assert (t2 == t).all()
    assert int((keys != 0).sum().item()) == len(cloud)


def test_hash_generic_keys(ops):
    rng = np.random.default_rng(0)
    k = np.unique(rng.integers(1, 1 << 40, 5000))
    keys = torch.zeros(2 * len(k), dtype=torch.int64, device='cuda')
    vals = torch.zeros(2 * len(k), dtype=torch.int32, device='cuda')
    ops.hash_insert_keys(keys, vals, _cuda(k))
    probe = np.concatenate((k[::3], rng.integers(1 << 41, 1 << 42, 100)))
    got = ops.hash_lookup_keys(keys, vals, _cuda(probe)).cpu().numpy()[:len(probe)]
    want = np.concatenate((np.arange(len(k))[::3] + 1, np.zeros(100, np.int64)))
    assert (got == want).all()


SHAPES = [(256, 256, (3, 3, 3), (1, 1, 1)), (1, 64, (3, 3, 3), (1, 1, 1)), (8, 32, (2, 2, 2), (2, 2, 2)),
          (8, 64, (4, 4, 4), (4, 4, 4)), (264, 96, (1, 1, 1), (1, 1, 1)), (64, 255, (1, 1, 1), (1, 1, 1)),
          (32, 320, (3, 3, 3), (1, 1, 1))]


@pytest.mark.parametrize('c_in,c_out,ks,st', SHAPES)
@pytest.mark.parametrize('mode', ['raw', 'i8_prelu', 'i32_bias'])
def test_conv_i8_bit_exact(ops, cloud, c_in, c_out, ks, st, mode):
    rng = np.random.default_rng(c_in * 7 + c_out)
    out_c = cloud.copy()
    if st != (1, 1, 1):
        out_c[:, 1:] >>= st[0].bit_length() - 1
        out_c = np.unique(out_c, axis=0)
    volume = ks[0] * ks[1] * ks[2]
    a = rng.integers(-127, 128, (len(cloud), c_in)).astype(np.int8)
    w = rng.integers(-127, 128, (volume, c_out, c_in)).astype(np.int8)
    comp = rng.integers(-5000, 5000, (volume, c_out)).astype(np.int32) if mode == 'i32_bias' else None
    bias = rng.integers(-20000, 20000, c_out).astype(np.int32)
    mul = rng.integers(1 << 10, 1 << 18, c_out).astype(np.int64)
    slope = np.array([int(0.3 * (1 << 25))], dtype=np.int32)
    zp = np.array([12345], dtype=np.int64)
    table = oi.kernel_table(cloud, out_c, ks, st)
    acc = oi.conv_i8(a, table if volume > 1 else None, w, comp)
    if volume > 1:
        _, _, t = _table_gpu(ops, cloud, out_c, ks, st)
        kw = dict(nbr=t, n_offsets=volume, nbr_ks=1, nbr_os=volume, nbr_bias=1)
    else:
        kw = {}
    from fastpcc_amd.int_sparse_conv import _pad_weight
    wd = _pad_weight(_cuda(w))
    if mode == 'raw':
        got = ops.conv_i8(_cuda(a), wd, c_in, c_out, len(out_c), **kw)
        want = acc
    elif mode == 'i8_prelu':
        got = ops.conv_i8(_cuda(a), wd, c_in, c_out, len(out_c), bias=_cuda(bias), slope=_cuda(slope),
                          requant_mul=_cuda(mul), zero_point=_cuda(zp), shift=20, out_bits=8, **kw)
        want = oi.epilogue(acc, bias, slope, mul, 12345, 20, 8).astype(np.int8)
    else:
        got = ops.conv_i8(_cuda(a), wd, c_in, c_out, len(out_c), zp_comp=_cuda(comp), bias=_cuda(bias),
                          requant_mul=_cuda(mul).to(torch.uint32), zero_point=_cuda(zp), shift=3, out_bits=32, **kw)
        want = oi.epilogue(acc, bias, None, mul, 12345, 3, 32)
    assert got.dtype == (torch.int8 if mode == 'i8_prelu' else torch.int32)
    assert (got.cpu().numpy() == want).all()


def test_standalone_epilogues(ops):
    rng = np.random.default_rng(2)
    x = rng.integers(-2 ** 31, 2 ** 31, (1000, 264)).astype(np.int32)
    x[0, :8] = [2 ** 31 - 1, -2 ** 31, 0, 1, -1, 3, -3, 12345]
    slope = np.array([int(-0.7 * (1 << 25))], dtype=np.int32)
    # RequantFxpToScaledInt8: one multiplier, shift 23 + s
    mul1 = np.array([1017], dtype=np.int64)
    got = ops.epilogue_i32(_cuda(x), _cuda(mul1).to(torch.uint32), _cuda(np.array([0], np.int64)), 23 + 5, 8)
    assert (got.cpu().numpy() == oi.epilogue(x, None, None, mul1, 0, 28, 8).astype(np.int8)).all()
    mulc = rng.integers(1, 2 ** 32, 264).astype(np.int64)                       # full uint32 range, as int64 in checkpoints
    b = rng.integers(-2 ** 20, 2 ** 20, 264).astype(np.int32)
    got = ops.epilogue_i32(_cuda(x), _cuda(mulc), _cuda(np.array([-77], np.int64)), 31, 32, bias=_cuda(b), slope=_cuda(slope))
    assert (got.cpu().numpy() == oi.epilogue(x, b, slope, mulc, -77, 31, 32)).all()
    # prelu with and without the fused (wrapping) residual add
    y = rng.integers(-2 ** 31, 2 ** 31, x.shape).astype(np.int32)
    got = ops.prelu_i32(_cuda(x), _cuda(slope)).cpu().numpy()
    assert (got == oi.prelu_i32(x, int(slope[0]))).all()
    got = ops.prelu_i32(_cuda(x), _cuda(slope), add=_cuda(y)).cpu().numpy()
    s = (x.astype(np.int64) + y.astype(np.int64)).astype(np.int32)
    assert (got == oi.prelu_i32(s, int(slope[0]))).all()


def test_grouped_epilogue_and_occupied_octant_linear(ops):
    """epilogue with one parameter set per row group, and the linear layer C -> 8C evaluated for the occupied (row, octant)
    pairs only (an 8-offset gather convolution with one table entry per output row) against the dense evaluation"""
    rng = np.random.default_rng(11)
    n, c_in, ch = 3000, 64, 48
    x = rng.integers(-127, 128, (n, c_in)).astype(np.int8)
    w = rng.integers(-127, 128, (8 * ch, c_in)).astype(np.int8)
    bias = rng.integers(-30000, 30000, 8 * ch).astype(np.int32)
    mul = rng.integers(1 << 8, 1 << 20, 8 * ch).astype(np.int64)
    bits = rng.random((n, 8)) < 0.3
    bits[np.arange(n), rng.integers(0, 8, n)] = True
    dense = oi.epilogue(oi.conv_i8(x, None, w[None], None), bias, None, mul, 0, 9, 32).reshape(n, 8, ch)[bits]
    rows, octs = np.nonzero(bits)
    m = len(rows)
    table = np.zeros(((m + 127) // 128 * 128, 8), np.int32)
    table[np.arange(m), octs] = rows + 1
    from fastpcc_amd.int_sparse_conv import _pad_weight
    wd = _pad_weight(_cuda(w.reshape(8, ch, c_in)))
    raw = ops.conv_i8(_cuda(x), wd, c_in, ch, m, nbr=_cuda(table), n_offsets=8, nbr_ks=1, nbr_os=8, nbr_bias=1)
    got = ops.epilogue_i32(raw, _cuda(mul), _cuda(np.array([0], np.int64)), 9, 32, bias=_cuda(bias),
                           row_group=_cuda(octs.astype(np.int32)))
    assert (got.cpu().numpy() == dense).all()
    with pytest.raises(ValueError):
        ops.epilogue_i32(raw, _cuda(mul[:ch + 1]), None, 9, 32, row_group=_cuda(octs.astype(np.int32)))


def test_conv_i8_row_order_changes_nothing(ops, cloud):
    rng = np.random.default_rng(5)
    n = len(cloud)
    a = _cuda(rng.integers(-127, 128, (n, 64)).astype(np.int8))
    from fastpcc_amd.int_sparse_conv import _pad_weight
    w = _pad_weight(_cuda(rng.integers(-127, 128, (27, 128, 64)).astype(np.int8)))
    _, _, t = _table_gpu(ops, cloud, cloud, (3, 3, 3), (1, 1, 1))
    order = ops.conv_row_order((t - 1).contiguous(), 27, 1, 27, n, 17)
    assert sorted(order.cpu().tolist()) == list(range(n))
    kw = dict(nbr=t, n_offsets=27, nbr_ks=1, nbr_os=27, nbr_bias=1)
    plain = ops.conv_i8(a, w, 64, 128, n, **kw)
    assert torch.equal(ops.conv_i8(a, w, 64, 128, n, row_order=order, **kw), plain)


@pytest.mark.parametrize('c', [255, 256, 2, 17])
def test_softmax_and_cdf(ops, c):
    rng = np.random.default_rng(c)
    x = (rng.normal(0, 3, (3001, c)) * (1 << 23)).astype(np.int64).clip(-2 ** 31, 2 ** 31 - 1).astype(np.int32)
    x[0] = 0
    x[1] = -2 ** 31
    x[2, 0] = 2 ** 31 - 1
    q = (x >> 7).astype(np.int32)
    got = ops.softmax_i32(_cuda(q)).cpu().numpy().astype(np.int64) & 0xffffffff
    assert (got == oi.softmax_i32(q).astype(np.int64)).all()
    cdf = ops.logits_to_cdf16(_cuda(x), 7).cpu().numpy().view(np.uint16)
    want = oi.quantize_pmf(x)
    assert (cdf == want).all()
    sym = rng.integers(0, c, len(x)).astype(np.int16)
    start, fm1 = ops.logits_to_ranges(_cuda(x), 7, _cuda(sym))
    w64 = want.astype(np.int64)
    lo = np.where(sym > 0, w64[np.arange(len(x)), np.maximum(sym - 1, 0)], 0)
    hi = np.where(sym == c - 1, 65536, w64[np.arange(len(x)), sym])
    assert (start.cpu().numpy().view(np.uint16) == lo).all()
    assert (fm1.cpu().numpy().view(np.uint16) == hi - lo - 1).all()


@pytest.mark.parametrize('shift', [1, 20, 31, 32, 33, 40, 60])
@pytest.mark.parametrize('case', ['plain', 'slope_one', 'slope_big', 'slope_negative_one', 'bias_huge', 'mul_huge', 'no_prelu'])
@pytest.mark.parametrize('out_bits', [8, 32])
def test_tiled_epilogue_forms_bit_exact(ops, cloud, shift, case, out_bits):
    """the fused epilogue of the tiled int8 convolution picks between a form without range checks (bounded accumulator, bias with room,
    slope in (-1, 1], multiplier < 2^31; a high-word shortcut for shifts >= 32), the checked form and the generic wrapping form: every
    combination equals the reference arithmetic (oracle/int_ops.c) exactly"""
    rng = np.random.default_rng(shift * 31 + len(case) + out_bits)
    c_in = c_out = 128
    n = len(cloud)
    a = rng.integers(-127, 128, (n, c_in)).astype(np.int8)
    w = rng.integers(-127, 128, (27, c_out, c_in)).astype(np.int8)
    bias = rng.integers(-20000, 20000, c_out).astype(np.int32)
    mul = rng.integers(1 << 10, 1 << 18, c_out).astype(np.int64)
    slope = np.array([int(0.3 * (1 << 25))], dtype=np.int32)
    if case == 'slope_one':
        slope[0] = 1 << 25
    elif case == 'slope_big':
        slope[0] = 3 << 25                                                  # PReLU may grow a value: checked form
    elif case == 'slope_negative_one':
        slope[0] = -(1 << 25)                                               # the excluded end of the slope interval
    elif case == 'bias_huge':
        bias[::3] = 2 ** 31 - 5                                             # acc + bias can leave int32: generic form for those waves
        bias[1::3] = -2 ** 31 + 7
    elif case == 'mul_huge':
        mul[::2] = (1 << 31) + 12345                                        # multipliers of 2^31 and more
    use_slope = case != 'no_prelu'
    zp = np.array([(12345 << min(shift, 40)) + 77], dtype=np.int64)
    table = oi.kernel_table(cloud, cloud, (3, 3, 3), (1, 1, 1))
    acc = oi.conv_i8(a, table, w, None)
    _, _, t = _table_gpu(ops, cloud, cloud, (3, 3, 3), (1, 1, 1))
    from fastpcc_amd.int_sparse_conv import _pad_weight
    got = ops.conv_i8(_cuda(a), _pad_weight(_cuda(w)), c_in, c_out, n, nbr=t, n_offsets=27, nbr_ks=1, nbr_os=27, nbr_bias=1,
                      bias=_cuda(bias), slope=_cuda(slope) if use_slope else None, requant_mul=_cuda(mul).to(torch.uint32) if case != 'mul_huge' else _cuda(mul),
                      zero_point=_cuda(zp), shift=shift, out_bits=out_bits)
    want = oi.epilogue(acc, bias, slope if use_slope else None, mul, int(zp[0]), shift, out_bits)
    assert (got.cpu().numpy() == (want.astype(np.int8) if out_bits == 8 else want)).all()
