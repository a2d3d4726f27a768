"""Self-consistency of the float oracle (MinkowskiEngine is not available): the two evaluations
of the convolution sum agree, kernel maps obey the restated ME semantics, and the oracle codec round-trips."""
import numpy as np
import pytest
import torch

from fastpcc_amd.codecs.lossy_coord_v2 import Model
from fastpcc_amd.codecs.lossy_coord_v2.model_config import baseline_r1
from oracle import coords as oc
from oracle import sparse_conv as sc
from oracle.codec_v2 import OracleV2
from util import batched, enliven, surface_cloud


def test_kernel_offsets_order():
    # first axis fastest, odd kernels centred, even kernels anchored at 0 (semantics (ii))
    o3 = oc.kernel_offsets(3, 2)
    assert o3[0].tolist() == [-2, -2, -2] and o3[1].tolist() == [0, -2, -2] and o3[13].tolist() == [0, 0, 0]
    assert o3[26].tolist() == [2, 2, 2] and o3[3].tolist() == [-2, 0, -2]
    # even kernels: the child table the reference itself states (minkowski_expand_coord_2x), from the generated fixture
    import json, os
    with open(os.path.join(os.path.dirname(__file__), 'golden', 'me_semantics.json')) as f:
        table = json.load(f)['expand_coord_2x']['2']
    assert oc.kernel_offsets(2, 1).tolist() == [row[1:] for row in table]


def test_levels_and_maps():
    xyz = surface_cloud(3, 32, 4000)
    lvl = oc.Level(batched(xyz), 1)
    assert lvl.n == len(xyz)
    assert (np.diff(oc.morton_encode(lvl.coords[:, 1:])) > 0).all()          # Morton order, unique
    up = oc.strided(lvl)
    assert (up.coords[:, 1:] % 2 == 0).all()
    km = oc.kernel_map(lvl, up, 2)
    assert sum(len(i) for i, _ in km) == lvl.n                                 # every voxel has exactly one parent
    for k, (rows_in, rows_out) in enumerate(km):
        d = lvl.coords[rows_in, 1:] - up.coords[rows_out, 1:]
        assert (d == oc.kernel_offsets(2, 1)[k]).all()
    gen = oc.generated(up)
    assert gen.n == 8 * up.n
    # generated rows are (parent, octant) ordered: row 8p+k = parent + offset_k
    assert (gen.coords.reshape(up.n, 8, 4)[:, :, 1:] - up.coords[:, None, 1:] == oc.kernel_offsets(2, 1)[None]).all()
    k3 = oc.kernel_map(lvl, lvl, 3)
    assert (k3[13][0] == k3[13][1]).all() and len(k3[13][0]) == lvl.n
    for k in range(27):                                                         # symmetry of the neighbourhood relation
        a = set(zip(k3[k][0].tolist(), k3[k][1].tolist()))
        b = set(zip(k3[26 - k][1].tolist(), k3[26 - k][0].tolist()))
        assert a == b


def test_conv_mm_equals_chain():
    rng = np.random.default_rng(0)
    xyz = surface_cloud(5, 32, 3000)
    lvl = oc.Level(batched(xyz), 1)
    kmap = oc.kernel_map(lvl, lvl, 3)
    for c_in, c_out in ((1, 16), (16, 8), (32, 32)):
        x = rng.normal(size=(lvl.n, c_in)).astype(np.float32)
        w = (rng.normal(size=(27, c_in, c_out)) / np.sqrt(13 * c_in)).astype(np.float32)
        b = rng.normal(size=c_out).astype(np.float32)
        ref = sc.conv_mm(torch.from_numpy(x), kmap, torch.from_numpy(w), torch.from_numpy(b), lvl.n, sc.ACT_PRELU, 0.25).numpy()
        for order in (0, 1) if c_in % 8 == 0 else (0,):
            got = sc.conv_chain_kmap(x, kmap, w, b, lvl.n, act=sc.ACT_PRELU, slope=0.25, order=order)
            np.testing.assert_allclose(got, ref, rtol=2e-5, atol=2e-5)
    # two-source input == concatenated input, bit for bit
    x1 = rng.normal(size=(lvl.n, 8)).astype(np.float32)
    x2 = rng.normal(size=(lvl.n, 8)).astype(np.float32)
    w = rng.normal(size=(27, 16, 4)).astype(np.float32)
    a = sc.conv_chain_kmap(np.concatenate((x1, x2), 1), kmap, w, None, lvl.n)
    b = sc.conv_chain_kmap(x1, kmap, w, None, lvl.n, x2=x2)
    assert (a == b).all()


def test_oracle_codec_roundtrip_and_modes_agree():
    cfg = baseline_r1()
    torch.manual_seed(0)
    model = Model(cfg)
    enliven(model, 0)
    xyz = surface_cloud(1, 64, 6000)
    coords = batched(xyz) + np.array([0, 5, 0, 9])
    streams = {}
    for mode in ('mm', 'chain'):
        o = OracleV2(model.state_dict(), cfg, conv=mode)
        data = o.compress(coords)
        rec = o.decompress(data)
        streams[mode] = (data, o.symbols)
        assert rec.shape == (len(xyz), 3)                     # adaptive pruning returns exactly the coded point count
        assert rec.min(0).tolist() >= coords[:, 1:].min(0).tolist()
        # header: offsets then the point count
        lo = coords[:, 1:].min(0)
        assert [int.from_bytes(data[i:i + 2], 'little') for i in (0, 2, 4)] == lo.tolist()
        assert int.from_bytes(data[6:9], 'little') == len(xyz)
    # the two evaluations of the convolution sum differ only by fp32 rounding: symbol streams nearly identical
    ra, rb = streams['mm'][1]['residual'], streams['chain'][1]['residual']
    assert ra.shape == rb.shape and np.mean(ra != rb) < 0.02
    assert abs(len(streams['mm'][0]) - len(streams['chain'][0])) <= 0.02 * len(streams['mm'][0])


def test_d1_known_answers():
    """hand-derived: A = {(0,0,0), (4,0,0)}, B = {(1,0,0)}: A->B distances 1 and 9 (mse1 = 5), B->A distance 1 (mse2 = 1)"""
    from oracle import metrics as om
    a = np.array([[0, 0, 0], [4, 0, 0]])
    b = np.array([[1, 0, 0]])
    r = om.d1(a, b, 1024)
    assert r['mse1      (p2point)'] == 5.0 and r['mse2      (p2point)'] == 1.0 and r['mseF      (p2point)'] == 5.0
    assert r['mseF,PSNR (p2point)'] == pytest.approx(10 * np.log10(3 * 1023 ** 2 / 5.0), rel=1e-12)
    rng = np.random.default_rng(0)
    q, p = rng.integers(0, 50, (300, 3)), rng.integers(0, 50, (200, 3))
    assert (om.nn_dist2(q, p)[0] == om.brute_nn_dist2(q, p)).all()


def test_d2_and_hausdorff_known_answers():
    """hand-derived point-to-plane / Hausdorff values of oracle.metrics.d2 (pc_error's definitions with normals):
    (1) the plane z = 3 against z = 4: every error vector is the unit normal -> p2point = p2plane = h. = 1;
    (2) the same plane slid by one voxel inside itself: the border column is 1 away, but along the plane -> p2plane 0, h.(p2point) 1;
    (3) given normals: A = {0} with n = (0.6, 0.8, 0), B = {(5, 0, 0)}: p2point 25, both plane errors (5 * 0.6)^2 = 9;
    (4) a tie: A = {0}, n = (1, 0, 0); B = {(1,0,0), (0,1,0)} both at distance 1 -> A->B averages the projections 1 and 0 = 0.5; the first
        of B in Morton order receives A's normal, the other takes it from its nearest voxel of A: B->A = (1 + 0) / 2."""
    from oracle import metrics as om
    g = np.stack(np.meshgrid(np.arange(12), np.arange(12), indexing='ij'), -1).reshape(-1, 2)
    a = np.concatenate([g, np.full((len(g), 1), 3)], 1)
    r = om.d2(a, a + np.array([0, 0, 1]), 64, knn=9)
    for key in ('mse1      (p2plane)', 'mse2      (p2plane)', 'mseF      (p2plane)', 'h.        (p2point)', 'h.        (p2plane)'):
        assert r[key] == pytest.approx(1.0, abs=1e-12), key
    assert r['mseF,PSNR (p2plane)'] == pytest.approx(10 * np.log10(3 * 63 ** 2), rel=1e-12)
    r = om.d2(a, a + np.array([1, 0, 0]), 64, knn=9)
    assert r['mseF      (p2plane)'] == pytest.approx(0.0, abs=1e-12) and r['h.        (p2plane)'] == pytest.approx(0.0, abs=1e-12)
    assert r['h.        (p2point)'] == 1.0 and r['h.,PSNR   (p2plane)'] == float('inf')
    r = om.d2(np.array([[0, 0, 0]]), np.array([[5, 0, 0]]), 16, org_normals=np.array([[0.6, 0.8, 0.0]]))
    assert r['mse1      (p2plane)'] == pytest.approx(9.0, rel=1e-12) and r['mse2      (p2plane)'] == pytest.approx(9.0, rel=1e-12)
    assert r['h.       1(p2point)'] == 25.0 and r['h.       2(p2plane)'] == pytest.approx(9.0, rel=1e-12)
    r = om.d2(np.array([[0, 0, 0]]), np.array([[0, 1, 0], [1, 0, 0]]), 16, org_normals=np.array([[1.0, 0.0, 0.0]]))
    assert r['mse1      (p2plane)'] == pytest.approx(0.5, rel=1e-12) and r['mse2      (p2plane)'] == pytest.approx(0.5, rel=1e-12)
    assert r['h.        (p2plane)'] == pytest.approx(1.0, rel=1e-12) and r['h.        (p2point)'] == 1.0


def test_pca_normals_of_the_oracle_on_a_sphere():
    """voxelised sphere of radius 20: PCA normals over 30 neighbours point along the radius (within the voxelisation's few degrees),
    with the sign rule n . (1, sqrt 2, sqrt 5) > 0; a concentric sphere of radius 21 is ~1 away along them"""
    from oracle import metrics as om
    g = np.stack(np.meshgrid(*(np.arange(-24, 25),) * 3, indexing='ij'), -1).reshape(-1, 3)
    r = np.sqrt((g ** 2).sum(1))
    shell = g[np.abs(r - 20) < 0.5] + 24
    order = om.morton_rows(shell)
    pts = shell[order]
    n = om.pca_normals(pts, om.knn_rows(pts, pts, 30)[0])
    radial = (pts - 24) / np.linalg.norm(pts - 24, axis=1, keepdims=True)
    assert (np.abs((n * radial).sum(1)) > 0.97).all()
    assert (n @ np.array([1.0, np.sqrt(2.0), np.sqrt(5.0)]) > 0).all()
    outer = g[np.abs(r - 21) < 0.5] + 24
    res = om.d2(shell, outer, 64)
    assert 0.6 < res['mseF      (p2plane)'] < 1.4 and res['mseF      (p2plane)'] <= om.d1(shell, outer, 64)['mseF      (p2point)'] + 1e-9
