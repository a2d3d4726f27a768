"""Coordinate kernels of libfpcc_hip.so (through the C ABI) against the NumPy oracle: bit-exact integer work."""
import numpy as np
import pytest
import torch

from oracle import coords as oc
from util import batched, surface_cloud

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ops():
    from fastpcc_amd import hipops
    return hipops


def _dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda()


def test_morton_matches_golden_and_oracle(ops, golden_dir):
    import json, os
    with open(os.path.join(golden_dir, 'morton.json')) as f:
        g = json.load(f)
    xyz = np.array(g['xyz'], dtype=np.int64)
    d = _dev(xyz, torch.int32)
    cols = {'xyz': (0, 1, 2), 'zyx': (2, 1, 0), 'yxz': (1, 0, 2)}
    for tag, keys in g['keys'].items():
        order, inv = tag.split('|')
        c = cols[order][::-1] if int(inv) else cols[order]
        assert ops.morton3d_encode(d, c).cpu().tolist() == keys
    # strided view (columns 1..3 of a [n,4] tensor), as model code passes xyz[:, 1:]
    b = _dev(batched(xyz % 1024), torch.int32)
    assert ops.morton3d_encode(b[:, 1:]).cpu().tolist() == oc.morton_encode(xyz % 1024).tolist()
    assert ops.morton3d_encode(torch.zeros((0, 3), dtype=torch.int32, device='cuda')).numel() == 0


@pytest.mark.parametrize('res,n', [(64, 20000), (256, 200000)])
def test_pyramid_and_neighbour_tables(ops, res, n):
    rng = np.random.default_rng(res)
    xyz = surface_cloud(res, res, n)
    shuffled = xyz[rng.permutation(len(xyz))]
    coords = np.concatenate((batched(shuffled), batched(shuffled[:100])))       # with duplicates
    bits = 21
    keys = ops.keys_from_coords(_dev(coords, torch.int32), 0, bits)
    skeys, perm = ops.sort_keys(keys)
    ukeys, first, count = ops.unique_keys(skeys)
    n_u = int(count.item())
    lvl = oc.Level(batched(xyz), 1)
    assert n_u == lvl.n
    ukeys = ukeys[:n_u].contiguous()
    assert ukeys.cpu().tolist() == oc.morton_encode(lvl.coords[:, 1:]).tolist()
    # perm/first give an input row with the right coordinates
    rows = perm[first[:n_u].long()].cpu().numpy()
    assert (coords[rows] == lvl.coords).all()
    back = ops.coords_from_keys(ukeys, 0, bits).cpu().numpy()
    assert (back == lvl.coords).all()

    # --- pyramid: coarsen until small, compare every level with the oracle's strided maps --------------------------
    levels = [(ukeys, lvl, None, None)]
    cur_keys, cur_lvl = ukeys, lvl
    for l in range(1, 5):
        parent_of, pkeys, child_row, cnt = ops.coarsen(cur_keys)
        m = int(cnt.item())
        up = oc.strided(cur_lvl)
        assert m == up.n
        pkeys = pkeys[:m].contiguous()
        child_row = child_row[:m].contiguous()
        assert ops.coords_from_keys(pkeys, l, bits - l).cpu().numpy().tolist() == up.coords.tolist()
        km = oc.kernel_map(cur_lvl, up, 2)
        want_child = oc.dense_table(km, up.n).T                                   # [m, 8]
        assert (child_row.cpu().numpy() == want_child).all()
        want_parent = np.empty(cur_lvl.n, dtype=np.int64)
        for rows_in, rows_out in km:
            want_parent[rows_in] = rows_out
        assert (parent_of.cpu().numpy() == want_parent).all()
        levels.append((pkeys, up, parent_of, child_row))
        cur_keys, cur_lvl = pkeys, up

    # --- 27-neighbour tables: search at the top, parent-derived below; both must equal the oracle -------------------
    top_keys, top_lvl, _, _ = levels[-1]
    nbr = ops.nbr27_search(top_keys, bits - (len(levels) - 1))
    assert (nbr.cpu().numpy() == oc.dense_table(oc.kernel_map(top_lvl, top_lvl, 3), top_lvl.n)).all()
    for li in range(len(levels) - 2, -1, -1):
        keys_l, lvl_l, _, _ = levels[li]
        _, _, parent_of, child_row = levels[li + 1]
        parent_nbr = nbr
        nbr = ops.nbr27_from_parent(keys_l, parent_of, nbr, child_row)
        want = oc.dense_table(oc.kernel_map(lvl_l, lvl_l, 3), lvl_l.n)
        assert (nbr.cpu().numpy() == want).all()
        # the same pass with the row-major copy and the presence masks (round 6), and what is built on them
        nbr2, rows, masks = ops.nbr27_from_parent_ex(keys_l, parent_of, parent_nbr, child_row)
        assert torch.equal(nbr2, nbr)
        assert rows.shape == (lvl_l.n, 32) and torch.equal(rows[:, :27], nbr.t()) and bool((rows[:, 27:] == -1).all())
        want_masks = ((want >= 0).astype(np.int64) << np.arange(27)[:, None]).sum(0)
        assert (masks.cpu().numpy().astype(np.int64) == want_masks).all()
        order_t = ops.conv_row_order(nbr, 27, lvl_l.n, 1, lvl_l.n, 13)
        order_m = ops.conv_row_order(None, 27, lvl_l.n, 1, lvl_l.n, 13, masks=masks)
        assert torch.equal(order_t, order_m)
        assert torch.equal(ops.gather_table_rows(rows, order_m), rows.index_select(0, order_m.long()))
        if lvl_l.n < 60000:
            assert (ops.nbr27_search(keys_l, bits - li).cpu().numpy() == want).all()

    # --- generated set of level 1 and its neighbour table ------------------------------------------------------------
    keys1, lvl1, _, _ = levels[1]
    nbr1 = ops.nbr27_search(keys1, bits - 1)
    gen = oc.generated(lvl1)
    got = ops.nbr27_from_parent(None, None, nbr1, None).cpu().numpy()
    assert (got == oc.dense_table(oc.kernel_map(gen, gen, 3), gen.n)).all()
    # the one-output-channel gather on the generated set straight from the PARENT's table (no table of the candidates' own): the bits of
    # fpcc_gather_sum_f32 on the built table
    y = torch.from_numpy(rng.normal(size=(gen.n, 32)).astype(np.float32)).cuda()
    bias, slope = torch.tensor([0.37], device='cuda'), torch.tensor([0.2], device='cuda')
    want_sum = ops.gather_sum(y, torch.from_numpy(got).cuda(), 27, gen.n, 1, gen.n, bias=bias, act=ops.ACT_PRELU, slope=slope, clip=3.0)
    got_sum = ops.gather_sum_generated(y, nbr1, bias=bias, act=ops.ACT_PRELU, slope=slope, clip=3.0)
    assert torch.equal(got_sum.view(torch.int32), want_sum.view(torch.int32))
    got2, rows_g, masks_g = ops.nbr27_from_parent_ex(None, None, nbr1, None)
    assert (got2.cpu().numpy() == got).all() and (rows_g[:, :27].t().cpu().numpy() == got).all() and bool((rows_g[:, 27:] == -1).all())
    assert (masks_g.cpu().numpy().astype(np.int64) == ((got >= 0).astype(np.int64) << np.arange(27)[:, None]).sum(0)).all()

    # --- refine: a random mask over the candidates -------------------------------------------------------------------
    mask = rng.random(8 * lvl1.n) < 0.4
    keys_r, parent_r, child_r, cnt = ops.refine(keys1, _dev(mask.astype(np.uint8)))
    n_r = int(cnt.item())
    sub = oc.Level(gen.coords[mask], gen.stride)
    assert n_r == sub.n == int(mask.sum())
    assert ops.coords_from_keys(keys_r[:n_r].contiguous(), 0, bits).cpu().numpy().tolist() == sub.coords.tolist()
    assert (parent_r[:n_r].cpu().numpy() == np.nonzero(mask)[0] // 8).all()
    want_child = np.where(mask, np.cumsum(mask) - 1, -1).reshape(-1, 8)
    assert (child_r.cpu().numpy() == want_child).all()
    nbr_r = ops.nbr27_from_parent(keys_r[:n_r].contiguous(), parent_r[:n_r].contiguous(), nbr1, child_r)
    assert (nbr_r.cpu().numpy() == oc.dense_table(oc.kernel_map(sub, sub, 3), sub.n)).all()
    xyz_r, cnt2 = ops.compact_coords(keys1, _dev(mask.astype(np.uint8)), 0, bits,
                                     _dev(np.array([7, 0, 3]), torch.int32))
    assert int(cnt2.item()) == n_r
    assert (xyz_r[:n_r].cpu().numpy() == sub.coords[:, 1:] + np.array([7, 0, 3])).all()


def test_batched_clouds_do_not_mix(ops):
    a, b = surface_cloud(11, 32, 3000), surface_cloud(12, 32, 3000)
    coords = np.concatenate((batched(a, 0), batched(b, 1)))
    bits = 20
    keys = ops.keys_from_coords(_dev(coords, torch.int32), 0, bits)
    skeys, _ = ops.sort_keys(keys)
    lvl = oc.Level(coords, 1)
    assert ops.coords_from_keys(skeys, 0, bits).cpu().numpy().tolist() == lvl.coords.tolist()
    nbr = ops.nbr27_search(skeys, bits).cpu().numpy()
    assert (nbr == oc.dense_table(oc.kernel_map(lvl, lvl, 3), lvl.n)).all()


@pytest.mark.parametrize('clouds', [1, 3])
def test_level_counts_equal_the_coarsening_counts(ops, clouds):
    """fpcc_level_histogram: the rows of every coarser level from one pass over the finest keys = what level-by-level fpcc_coarsen
    counts (also across batch entries), and CoordinateManager.build_pyramid builds the same maps with it"""
    coords = np.concatenate([batched(surface_cloud(20 + b, 64, 5000), b) for b in range(clouds)])
    bits = 7
    keys, _ = ops.sort_keys(ops.keys_from_coords(_dev(coords, torch.int32), 0, bits))
    got = ops.level_counts(keys, bits - 1)
    want, cur = [], keys
    for _ in range(bits - 1):
        _, pkeys, _, cnt = ops.coarsen(cur)
        want.append(int(cnt.item()))
        cur = pkeys[:want[-1]].contiguous()
    assert got == want and want[-1] == clouds
    assert ops.level_counts(keys[:1], 3) == [1, 1, 1] and ops.level_counts(keys[:0], 2) == [0, 0]
    from fastpcc_amd import engine as ME
    cm = ME.CoordinateManager(D=3)
    key, _ = cm.insert_and_map(_dev(coords, torch.int32), 1)
    cm.build_pyramid(key, 4)
    m, rows = cm._map(key), []
    for _ in range(4):
        assert m.parent is not None
        m = m.parent
        rows.append(m.n)
    lvl, expect = oc.Level(coords, 1), []
    for _ in range(4):
        lvl = oc.strided(lvl)
        expect.append(lvl.n)
    assert rows == expect
    ME.clear_global_coordinate_manager()


def test_empty_and_tiny_inputs(ops):
    e = torch.zeros(0, dtype=torch.int64, device='cuda')
    s, p = ops.sort_keys(e)
    assert s.numel() == 0 and p.numel() == 0
    u, f, c = ops.unique_keys(e)
    assert int(c.item()) == 0
    one = ops.keys_from_coords(torch.tensor([[0, 5, 6, 7]], dtype=torch.int32, device='cuda'), 0, 21)
    par, pk, cr, cnt = ops.coarsen(one)
    assert int(cnt.item()) == 1 and par.cpu().tolist() == [0]
    assert cr[0].cpu().tolist() == [-1, -1, -1, -1, -1, 0, -1, -1]      # (5,6,7): octant = (x&1) + 2(y&1) + 4(z&1) = 5


def test_hilbert_keys_match_reference_golden_and_oracle():
    import json, os
    from fastpcc_amd import hipops as ops
    from oracle.hilbert import hilbert3d_encode
    with open(os.path.join(os.path.dirname(__file__), 'golden', 'hilbert.json')) as f:
        G = json.load(f)
    for case in G:
        xyz = torch.tensor(case['xyz'], dtype=torch.int32).cuda()
        assert ops.hilbert3d_encode(xyz, case['bits'], case['cols']).cpu().tolist() == case['keys']
    rng = np.random.default_rng(3)
    big = rng.integers(0, 1 << 21, (200000, 4)).astype(np.int32)           # a wider row: columns 1..3 hold the axes
    got = ops.hilbert3d_encode(torch.from_numpy(big).cuda(), 21, (1, 2, 3)).cpu().numpy()
    assert (got == hilbert3d_encode(big, 21, (1, 2, 3))).all()
    assert ops.hilbert3d_encode(torch.zeros((0, 3), dtype=torch.int32, device='cuda'), 10).shape == (0,)
