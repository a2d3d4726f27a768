"""fpcc_nn_dist2 / fastpcc_amd.evaluators (D1 distortion on the device) against the CPU oracle: integer squared distances
must agree exactly."""
import numpy as np
import pytest
import torch

from oracle import metrics as om
from util import surface_cloud

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ops():
    from fastpcc_amd import hipops
    return hipops


def _keys(ops, pts, bits, batch=None):
    c = np.zeros((len(pts), 4), np.int32)
    c[:, 1:] = pts
    if batch is not None:
        c[:, 0] = batch
    t = torch.from_numpy(c).cuda()
    keys, _ = ops.sort_keys(ops.keys_from_coords(t, 0, bits), 63)
    return t, keys


@pytest.mark.parametrize('seed,jitter', [(0, 1), (1, 3), (2, 40)])
def test_nn_dist2_matches_kdtree(ops, seed, jitter):
    rng = np.random.default_rng(seed)
    a = surface_cloud(seed, 128, 30000)
    b = np.unique(np.clip(a + rng.integers(-jitter, jitter + 1, a.shape), 0, 127), axis=0)[: len(a) - 500]
    qa, _ = _keys(ops, a, 7)
    _, kb = _keys(ops, b, 7)
    d, rows = ops.nn_dist2(kb, 7, qa, want_rows=True)
    want, _ = om.nn_dist2(a, b)
    assert (d.cpu().numpy() == want).all()
    # the reported row really is at that distance
    kb_xyz = ops.coords_from_keys(kb, 0, 7).cpu().numpy()[:, 1:]
    got_rows = rows.cpu().numpy()
    assert (((a.astype(np.int64) - kb_xyz[got_rows]) ** 2).sum(1) == want).all()


def test_far_sparse_and_edge_queries(ops):
    pts = np.array([[0, 0, 0], [1023, 1023, 1023], [512, 3, 900]], dtype=np.int64)
    q = np.array([[0, 0, 0], [1023, 0, 0], [500, 500, 500], [1, 1, 1], [1023, 1023, 1022], [0, 1023, 0], [700, 2, 901]], dtype=np.int64)
    qa, _ = _keys(ops, q, 10)
    _, kb = _keys(ops, pts, 10)
    d = ops.nn_dist2(kb, 10, qa).cpu().numpy()
    assert (d == om.brute_nn_dist2(q, pts)).all()


def test_batches_do_not_see_each_other_and_empty_set(ops):
    pts = np.array([[5, 5, 5], [6, 5, 5]], dtype=np.int64)
    _, kb = _keys(ops, pts, 4, batch=np.array([0, 1]))
    q = np.array([[6, 5, 5], [6, 5, 5], [0, 0, 0]], dtype=np.int64)
    qa, _ = _keys(ops, q, 4, batch=np.array([0, 1, 2]))
    d = ops.nn_dist2(kb, 4, qa).cpu().numpy()
    assert d.tolist() == [1, 0, -1]
    empty = torch.empty(0, dtype=torch.int64, device='cuda')
    assert ops.nn_dist2(empty, 4, qa).cpu().numpy().tolist() == [-1, -1, -1]
    assert int(ops.sum_i64(ops.nn_dist2(kb, 4, qa)).item()) == 1        # negative entries are not summed


def test_evaluator_matches_oracle_and_reference_keys():
    from fastpcc_amd.evaluators import PCCEvaluator, d1_metrics
    rng = np.random.default_rng(4)
    org = surface_cloud(4, 256, 80000)
    rec = np.unique(np.clip(org + rng.integers(-2, 3, org.shape), 0, 255), axis=0)
    rec = rec[rng.permutation(len(rec))[: len(org) - 1000]]
    got = d1_metrics(torch.from_numpy(org).cuda(), torch.from_numpy(rec).cuda(), 256)
    want = om.d1(org, rec, 256)
    for k, v in want.items():
        assert got[k] == pytest.approx(v, rel=1e-12), k
    assert got['mse1+mse2 (p2point)'] == got['mse1      (p2point)'] + got['mse2      (p2point)']
    same = d1_metrics(torch.from_numpy(org).cuda(), torch.from_numpy(org).cuda(), 256)
    assert same['mseF      (p2point)'] == 0 and same['mseF,PSNR (p2point)'] == float('inf')
    ev = PCCEvaluator()
    for name in ('a.ply', 'b.ply'):
        ev.log(torch.from_numpy(rec).cuda(), len(org), b'x' * 1000, name, 256, org_xyz=torch.from_numpy(org).cuda())
    mean = ev.show(None)
    assert mean['samples_num'] == 2 and mean['bpp(mean)'] == pytest.approx(8000 / len(org))
    assert mean['mseF,PSNR (p2point)(mean)'] == pytest.approx(want['mseF,PSNR (p2point)'], rel=1e-12)


def test_colour_psnr_of_identical_clouds_is_infinite():
    from fastpcc_amd.evaluators import d1_metrics
    org = surface_cloud(6, 64, 5000)
    col = torch.randint(0, 256, (len(org), 3), device='cuda')
    perm = torch.randperm(len(org), device='cuda')
    xyz = torch.from_numpy(org).cuda()
    out = d1_metrics(xyz, xyz[perm], 64, col, col[perm])
    assert out['c[0],PSNRF'] == float('inf') and out['c[3],PSNRF'] == float('inf')
    noisy = (col[perm] + 2).clamp(max=255)
    out = d1_metrics(xyz, xyz[perm], 64, col, noisy)
    assert 35 < out['c[0],PSNRF'] < 50


@pytest.mark.parametrize('K', [1, 3, 8, 16])
def test_knn3d_matches_a_dense_distance_matrix(K):
    """fpcc_knn3d (lib.knn3d.knn3d's entry): the K smallest squared distances per query equal those of torch.cdist, and
    every returned index really is at the returned distance"""
    from fastpcc_amd import hipops as ops
    g = torch.Generator().manual_seed(K)
    p1 = (torch.rand((3000, 3), generator=g) * 100).cuda()
    p2 = (torch.rand((2500, 3), generator=g) * 100).cuda()
    idx, d2 = ops.knn3d(p1, p2, K)
    ref = (torch.cdist(p1.double(), p2.double()) ** 2).topk(K, dim=1, largest=False).values
    assert idx.shape == (3000, K) and idx.dtype == torch.int64 and (idx >= 0).all() and (idx < 2500).all()
    assert torch.allclose(d2.double(), ref, rtol=1e-5, atol=1e-3)
    assert (d2[:, 1:] >= d2[:, :-1]).all()
    at = ((p1[:, None, :] - p2[idx]) ** 2).sum(-1)
    assert torch.allclose(at, d2, rtol=1e-5, atol=1e-3)
    few_idx, few_d = ops.knn3d(p1[:5], p2[:2], 3)                       # fewer candidates than K: -1 / inf fill
    assert (few_idx[:, 2] == -1).all() and torch.isinf(few_d[:, 2]).all() and (few_idx[:, :2] >= 0).all()


def test_knn3d_takes_integer_and_strided_inputs():
    """inputs that need a conversion copy (int32 voxel coordinates, a non-contiguous view): the copies must stay alive until
    the launch -- same sizes on both sides make the caching allocator most likely to hand one block out twice otherwise"""
    from fastpcc_amd import hipops as ops
    g = torch.Generator().manual_seed(5)
    a = torch.randint(0, 200, (2000, 3), generator=g, dtype=torch.int32).cuda()
    b = torch.randint(0, 200, (2000, 3), generator=g, dtype=torch.int32).cuda()
    idx, d2 = ops.knn3d(a, b, 4)
    exact = lambda p, q, k: ((p[:, None, :].long() - q[None, :, :].long()) ** 2).sum(-1).topk(k, dim=1, largest=False).values
    assert torch.equal(d2.long(), exact(a, b, 4))                           # integer coordinates: exact
    wide = torch.randint(0, 200, (2000, 6), generator=g, dtype=torch.int32).cuda()
    idx2, d3 = ops.knn3d(wide[:, ::2], wide[:, 1::2], 2)
    assert torch.equal(d3.long(), exact(wide[:, ::2], wide[:, 1::2], 2))


@pytest.mark.gpu
def test_clock_probe_reports_a_plausible_shader_clock():
    """fpcc_clock_probe (diagnostic): shader cycles per 100 MHz tick of a 200-us spin lie in a GPU's range"""
    from fastpcc_amd import hipops
    out = torch.zeros(2, dtype=torch.int64, device='cuda')
    hipops.clock_probe(out, 200)
    torch.cuda.synchronize()
    cycles, ticks = out.tolist()
    assert 200 * 100 <= ticks < 400 * 100
    assert 500 < cycles / ticks * 100 < 3500
    with pytest.raises(Exception):
        hipops.clock_probe(out, 0)


# ------------------------------------------------------------------------------------------------------------------------------
# point-to-plane (D2) and Hausdorff: fpcc_knn_voxels / fpcc_pca_normals / fpcc_transfer_normals / fpcc_nn_plane_dist2 against
# oracle/metrics.py (whose hand-derived answers are in tests/test_oracle_float.py)

def _sorted_cloud(ops, pts, bits):
    t, keys = _keys(ops, pts, bits)
    order = om.morton_rows(pts)
    srt = pts[order]
    q = np.zeros((len(srt), 4), np.int32)
    q[:, 1:] = srt
    return srt, torch.from_numpy(q).cuda(), keys


@pytest.mark.parametrize('k,start', [(1, 0), (8, 0), (9, 2), (16, 1), (30, 1), (32, 4)])
def test_knn_voxels_matches_the_oracle_order(ops, k, start):
    """(squared distance, row) is a total order: rows and distances must be IDENTICAL, ties included; the start level is a hint"""
    pts = surface_cloud(11, 64, 3000)
    srt, q, keys = _sorted_cloud(ops, pts, 6)
    rows, d = ops.knn_voxels(keys, 6, q, k, start)
    want_rows, want_d = om.knn_rows(srt, srt, k)
    assert (d.cpu().numpy() == want_d).all()
    assert (rows.cpu().numpy() == want_rows).all()
    assert (rows[:, 0].cpu().numpy() == np.arange(len(srt))).all()          # a voxel is its own nearest neighbour
    # queries that are not voxels of the set, some outside the cube
    rng = np.random.default_rng(k)
    other = np.zeros((500, 4), np.int32)
    other[:, 1:] = rng.integers(-5, 70, (500, 3))
    rows, d = ops.knn_voxels(keys, 6, torch.from_numpy(other).cuda(), k, start)
    want_rows, want_d = om.knn_rows(other[:, 1:], srt, k)
    assert (d.cpu().numpy() == want_d).all() and (rows.cpu().numpy() == want_rows).all()


def test_knn_voxels_with_fewer_voxels_than_k(ops):
    pts = np.array([[1, 1, 1], [2, 1, 1], [9, 9, 9]])
    srt, q, keys = _sorted_cloud(ops, pts, 4)
    rows, d = ops.knn_voxels(keys, 4, q, 5)
    assert (rows[:, 3:].cpu().numpy() == -1).all() and (d[:, 3:].cpu().numpy() == -1).all()
    assert sorted(rows[0, :3].tolist()) == [0, 1, 2]


def test_pca_normals_match_numpy_where_the_data_determine_them(ops):
    pts = surface_cloud(12, 128, 20000)
    srt, q, keys = _sorted_cloud(ops, pts, 7)
    rows, _ = ops.knn_voxels(keys, 7, q, 30)
    got = ops.pca_normals(keys, 7, rows).cpu().numpy()
    nbr = rows.cpu().numpy().astype(np.int64)
    want = om.pca_normals(srt, nbr)
    gap = om.eigen_gap(srt, nbr)
    ok = gap > 1e-6
    assert ok.mean() > 0.99
    assert np.abs(np.linalg.norm(got, axis=1) - 1).max() < 1e-12
    assert (got @ np.array([1.0, np.sqrt(2.0), np.sqrt(5.0)]) >= 0).all()
    # closed form against the iterative solver: the angle error scales with 1 / gap
    dots = (got * want).sum(1)
    assert (1 - dots[ok] < 1e-9 / gap[ok] ** 2 + 1e-12).all(), float((1 - dots[ok]).max())
    # degenerate inputs: fewer than three neighbours, a single repeated voxel
    few = torch.tensor([[0, 1, -1], [5, 5, 5]], dtype=torch.int32, device='cuda')
    assert ops.pca_normals(keys, 7, few).cpu().numpy().tolist() == [[0.0, 0.0, 1.0], [0.0, 0.0, 1.0]]


def _pair(seed):
    rng = np.random.default_rng(seed)
    org = surface_cloud(seed, 128, 12000)
    rec = np.unique(np.clip(org + rng.integers(-1, 2, org.shape), 0, 127), axis=0)
    rec = rec[rng.permutation(len(rec))[: len(org) - 500]]
    return org, rec


@pytest.mark.parametrize('seed', [3, 8])
def test_pc_error_metrics_with_given_normals_match_the_oracle(seed):
    """normals handed in (a PLY with nx ny nz): transfer to the reconstruction, tie-averaged projections, sums, maxima"""
    from fastpcc_amd.evaluators import pc_error_metrics
    org, rec = _pair(seed)
    rng = np.random.default_rng(seed + 100)
    normals = rng.normal(size=(len(org), 3))
    normals /= np.linalg.norm(normals, axis=1, keepdims=True)
    got = pc_error_metrics(torch.from_numpy(org).cuda(), torch.from_numpy(rec).cuda(), 128, org_normals=torch.from_numpy(normals).cuda(),
                           hausdorff=True)
    want = om.d2(org, rec, 128, org_normals=normals)
    assert set(want) <= set(got)
    for key, v in want.items():
        assert got[key] == pytest.approx(v, rel=1e-9, abs=1e-12), key
    for key, v in om.d1(org, rec, 128).items():
        assert got[key] == pytest.approx(v, rel=1e-12), key
    plain = pc_error_metrics(torch.from_numpy(org).cuda(), torch.from_numpy(rec).cuda(), 128, org_normals=torch.from_numpy(normals).cuda())
    assert not any(k.startswith('h.') for k in plain) and 'mseF,PSNR (p2plane)' in plain


def test_pc_error_metrics_with_estimated_normals_and_the_known_answers():
    from fastpcc_amd.evaluators import PCCEvaluator, estimate_normals, pc_error_metrics
    org, rec = _pair(5)
    got = pc_error_metrics(torch.from_numpy(org).cuda(), torch.from_numpy(rec).cuda(), 128, hausdorff=True)
    want = om.d2(org, rec, 128)
    for key, v in want.items():
        assert got[key] == pytest.approx(v, rel=1e-6, abs=1e-9), key          # PCA normals: closed form vs iterative solver
    assert got['mseF      (p2plane)'] <= got['mseF      (p2point)']
    # the estimate in the caller's row order
    n = estimate_normals(torch.from_numpy(org).cuda()).cpu().numpy()
    order = om.morton_rows(org)
    ref = om.pca_normals(org[order], om.knn_rows(org[order], org[order], 30)[0])
    assert np.median(np.abs((n[order] * ref).sum(1))) > 1 - 1e-9
    # tests/test_oracle_float.py's hand-derived cases on the device
    g = np.stack(np.meshgrid(np.arange(12), np.arange(12), indexing='ij'), -1).reshape(-1, 2)
    a = np.concatenate([g, np.full((len(g), 1), 3)], 1)
    dev = lambda x: torch.from_numpy(np.asarray(x)).cuda()
    r = pc_error_metrics(dev(a), dev(a + np.array([0, 0, 1])), 64, hausdorff=True, knn=9)
    for key in ('mse1      (p2plane)', 'mse2      (p2plane)', 'mseF      (p2plane)', 'h.        (p2point)', 'h.        (p2plane)'):
        assert r[key] == pytest.approx(1.0, abs=1e-12), key
    r = pc_error_metrics(dev(a), dev(a + np.array([1, 0, 0])), 64, hausdorff=True, knn=9)
    assert r['mseF      (p2plane)'] == pytest.approx(0.0, abs=1e-12) and r['h.        (p2point)'] == 1.0
    r = pc_error_metrics(dev([[0, 0, 0]]), dev([[5, 0, 0]]), 16, org_normals=dev([[0.6, 0.8, 0.0]]), hausdorff=True)
    assert r['mse1      (p2plane)'] == pytest.approx(9.0, rel=1e-9) and r['mse2      (p2plane)'] == pytest.approx(9.0, rel=1e-12)   # transferred normals: 2^-40 fixed point
    r = pc_error_metrics(dev([[0, 0, 0]]), dev([[0, 1, 0], [1, 0, 0]]), 16, org_normals=dev([[1.0, 0.0, 0.0]]), hausdorff=True)
    assert r['mse1      (p2plane)'] == pytest.approx(0.5, rel=1e-9) and r['mse2      (p2plane)'] == pytest.approx(0.5, rel=1e-12)
    # the evaluator logs the p2plane lines like a pc_error run that is given normals (the reference always gives them)
    ev = PCCEvaluator()
    ev.log(dev(rec), len(org), b'x' * 100, 'a.ply', 128, org_xyz=dev(org))
    mean = ev.show(None)
    assert mean['mseF,PSNR (p2plane)(mean)'] == pytest.approx(want['mseF,PSNR (p2plane)'], rel=1e-6)
    assert not any(k.startswith('h.') for k in mean)
    ev = PCCEvaluator(p2plane=False)
    ev.log(dev(rec), len(org), b'x' * 100, 'a.ply', 128, org_xyz=dev(org))
    assert not any('p2plane' in k for k in ev.show(None))


def test_sum_max_f64_is_reproducible_and_exact_on_integers(ops):
    rng = np.random.default_rng(0)
    for n in (0, 1, 255, 4096, 4097, 100000):
        v = rng.integers(0, 1000, n).astype(np.float64)
        got = ops.sum_max_f64(torch.from_numpy(v).cuda()).tolist()
        assert got[0] == v.sum() and (got[1] == (v.max() if n else -np.inf))
    v = torch.from_numpy(rng.normal(size=300000)).cuda()
    a, b = ops.sum_max_f64(v).tolist(), ops.sum_max_f64(v.clone()).tolist()
    assert a == b and a[0] == pytest.approx(float(v.sum()), rel=1e-9)
