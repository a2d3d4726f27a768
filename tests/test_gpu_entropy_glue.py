"""Small device kernels between the features and the coders, against NumPy/torch-CPU restatements of the reference lines."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ops():
    from fastpcc_amd import hipops
    return hipops


def test_logit_to_prob16(ops):
    rng = np.random.default_rng(0)
    x = np.concatenate((rng.normal(0, 4, 200000), [-100, 100, 0, -20, 20, 11.09, -11.09, np.inf, -np.inf])).astype(np.float32)
    got = ops.logit_to_prob16(torch.from_numpy(x).cuda()).cpu().numpy().view(np.uint16).astype(np.int64)
    # geo_lossl_em.py:95-99 on the CPU
    want = np.clip(np.round(torch.sigmoid(torch.from_numpy(x)).numpy().astype(np.float64) * 65536), 1, 65535).astype(np.int64)
    assert np.abs(got - want).max() <= 1                 # one ulp of fp32 sigmoid may move the rounded value by one
    assert np.mean(got != want) < 0.01
    assert got.min() >= 1 and got.max() <= 65535


def test_logit_to_prob16_is_the_specified_logistic_function_bit_for_bit(ops):
    """numerics version 3: the device kernel and oracle/sparse_conv.c evaluate the SAME sequence of fp32 operations (Cody-Waite
    reduction, degree-5 polynomial, explicit fused multiply-adds), so the 16-bit probabilities agree exactly -- on random logits, on a
    dense sweep across every binade the clamp leaves, and on the special values"""
    from oracle import sparse_conv as sc
    rng = np.random.default_rng(5)
    sweep = np.concatenate([s * np.ldexp(rng.uniform(1, 2, 4000), e) for e in range(-30, 8) for s in (-1.0, 1.0)])
    special = [0.0, -0.0, 87.0, -87.0, 87.5, -87.5, 88.8, -88.8, 1e30, -1e30, np.inf, -np.inf, 1e-45, -1e-45, 1.17549435e-38,
               0.5 * np.log(2), -0.5 * np.log(2), 11.0903, -11.0903, 16.6355, -16.6355]
    x = np.concatenate((rng.normal(0, 6, 1_000_000), rng.uniform(-90, 90, 200_000), sweep, special)).astype(np.float32)
    got = ops.logit_to_prob16(torch.from_numpy(x).cuda()).cpu().numpy().view(np.uint16).astype(np.int64)
    s = sc.sigmoid_spec(torch.from_numpy(x))
    want = np.clip(np.round(np.asarray(s, dtype=np.float64) * 65536), 1, 65535).astype(np.int64)
    assert (got == want).all(), int((got != want).sum())
    ref = torch.sigmoid(torch.from_numpy(x).double()).numpy()               # and it IS the logistic function: within 2 fp32 ulp
    assert np.abs(np.asarray(s, dtype=np.float64) - ref).max() <= 2.5e-7


def test_quantize_symbols(ops):
    rng = np.random.default_rng(1)
    x = np.concatenate((rng.uniform(-20, 20, 100000), [0.5, 1.5, 2.5, -0.5, -1.5, -2.5, 20, -20])).astype(np.float32)
    for scale in (1.0, 4.0):
        t = torch.from_numpy(x.copy()).cuda()
        sym = ops.quantize_symbols_(t, scale)
        r = torch.round(torch.from_numpy(x) * scale)       # half to even, like Tensor.round_()
        assert (sym.cpu() == r.to(torch.int32)).all()
        assert (t.cpu() == r / scale).all()


def test_child_mask(ops):
    cr = torch.tensor([[-1, 0, 1, -1, -1, 2, -1, -1], [3, -1, -1, -1, -1, -1, -1, 4]], dtype=torch.int32).cuda()
    assert ops.child_mask(cr).cpu().tolist() == [0, 1, 1, 0, 0, 1, 0, 0, 1, 0, 0, 0, 0, 0, 0, 1]


def _keep_reference(v: np.ndarray, target: int) -> np.ndarray:
    """Decoder.get_keep (lossy_coord_v2/layers.py:151-180) for one sample whose candidates are grouped by parent cell"""
    cells = v.reshape(-1, 8)
    not_max = (cells - cells.max(1, keepdims=True)) != 0
    ranked = np.sort(cells[not_max])
    thr = ranked[v.size - target - 1]
    return ((cells > thr) | ~not_max).reshape(-1)


@pytest.mark.parametrize('m,frac', [(1, 0.5), (37, 0.3), (5000, 0.26), (200000, 0.5)])
def test_topk_keep(ops, m, frac):
    rng = np.random.default_rng(m)
    v = rng.normal(size=8 * m).astype(np.float32)
    target = max(m, int(8 * m * frac))
    if target >= 8 * m:
        target = 8 * m - 1
    got = ops.topk_keep(torch.from_numpy(v).cuda(), target).cpu().numpy().astype(bool)
    want = _keep_reference(v, target)
    assert (got == want).all()
    assert got.sum() == target                                  # no ties in continuous random data


def test_topk_keep_with_ties(ops):
    v = np.array([1, 1, 0, 0, 0, 0, 0, 0, 3, 2, 2, 2, 1, 1, 1, 1], dtype=np.float32)
    for target in (3, 5, 9, 15):          # at least the three local maxima are always kept
        got = ops.topk_keep(torch.from_numpy(v).cuda(), target).cpu().numpy().astype(bool)
        assert (got == _keep_reference(v, target)).all()


@pytest.mark.parametrize('seed', [0, 1, 2])
def test_topk_keep_select_on_heavy_ties_and_every_sign(ops, seed):
    """the radix select behind fpcc_topk_keep: coarsely quantised logits (thousands of equal keys per bin, both signs, +-0, values that
    differ in the LAST key bits only) must give the mask of a full sort for any target, including 'keep everything'"""
    rng = np.random.default_rng(seed)
    m = 30000
    base = rng.normal(size=8 * m).astype(np.float32)
    v = [np.round(base * 4) / 4, np.where(rng.random(8 * m) < 0.5, np.float32(-0.0), np.float32(0.0)) + np.round(base),
         (np.float32(1.0) + rng.integers(0, 7, 8 * m).astype(np.float32) * np.float32(2.0 ** -23)) * np.sign(base)][seed].astype(np.float32)
    for target in (m, 8 * m // 3, 8 * m - 5, 8 * m):
        got = ops.topk_keep(torch.from_numpy(v).cuda(), target).cpu().numpy().astype(bool)
        if target == 8 * m:
            assert got.all()
        else:
            # with this many equal values a cell has several maxima: rank them at +inf like the kernel (and like sorting the
            # non-maxima, wherever the k-th value falls among those)
            cells = v.reshape(-1, 8)
            is_max = (cells == cells.max(1, keepdims=True)).reshape(-1)
            thr = np.sort(np.where(is_max, np.inf, v))[8 * m - target - 1]
            assert (got == ((v > thr) | is_max)).all(), target
            if 8 * m - target <= (~is_max).sum():
                assert (got == _keep_reference(v, target)).all(), target


def test_topk_keep_cells_select_matches_a_sort(ops):
    """fpcc_topk_keep_cells: cells of several candidate groups; threshold = the (8m - target)-th smallest of the candidates that are not
    their cell's maximum"""
    rng = np.random.default_rng(7)
    m, n_cells = 20000, 2500
    v = np.round(rng.normal(size=8 * m) * 8).astype(np.float32) / 8
    cell = np.sort(rng.integers(0, n_cells, m)).astype(np.int32)
    cell_max = np.full(n_cells, -np.inf, np.float32)
    np.maximum.at(cell_max, np.repeat(cell, 8), v)
    is_max = v == cell_max[np.repeat(cell, 8)]
    for target in (m, 3 * m, 8 * m - 1):
        ranked = np.sort(np.where(is_max, np.inf, v))
        thr = ranked[8 * m - target - 1]
        want = (v > thr) | is_max
        got = ops.topk_keep_cells(torch.from_numpy(v).cuda(), torch.from_numpy(cell).cuda(), n_cells, target).cpu().numpy().astype(bool)
        assert (got == want).all(), target
