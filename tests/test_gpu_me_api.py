"""The MinkowskiEngine names the reference's decoder reaches for directly -- MinkowskiMaxPooling, MinkowskiPoolingTranspose,
CoordinateMapKey look-ups by string id, decomposition_permutations, kernel_map, origin_map -- on fastpcc_amd.engine.

The expected keep masks are DATA: tests/golden/get_keep.json holds what the reference's `Decoder.get_keep`
(/root/reference/models/convolutional/lossy_coord_v2/layers.py:151-180) returned when tests/golden/make_golden.py executed it
on seeded candidate sets (coordinates, logits, requested point counts).  Here the same candidate sets are rebuilt on the
engine, and both the fused product kernel (fpcc_topk_keep) and a formulation through the engine's pooling operators must
reproduce those masks bit for bit."""
import json
import os

import numpy as np
import pytest
import torch

from util import batched, surface_cloud

pytestmark = pytest.mark.gpu

with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'get_keep.json')) as _f:
    CASES = {c['label']: c for c in json.load(_f)['cases']}


def _i16(hexstr):
    return np.frombuffer(bytes.fromhex(hexstr), dtype='<i2').reshape(-1, 4).astype(np.int64)


def _pack(c):
    """one sortable integer per (batch, x, y, z) row"""
    c = np.asarray(c, dtype=np.int64)
    return ((c[:, 0] << 48) | (c[:, 1] << 32) | (c[:, 2] << 16) | c[:, 3])


def _rebuild(ME, case):
    """the fixture's candidate set on fastpcc_amd.engine: stride-4 parents -> all children -> pruned to the fixture's stride-2
    voxels -> candidates at stride 1 carrying the fixture's logits (rows matched by coordinate, never by position)"""
    top_c = torch.from_numpy(_i16(case['top_coords_i16'])).to(torch.int32).cuda()
    cm = ME.CoordinateManager(D=3)
    top = ME.SparseTensor(torch.ones((top_c.shape[0], 1), device='cuda'), coordinates=top_c, tensor_stride=4, coordinate_manager=cm)
    up = ME.MinkowskiGenerativeConvolutionTranspose(1, 1, 2, 2, bias=False, dimension=3).cuda()
    with torch.no_grad():
        mid = up(top)                                                        # key (2, '')
        member = np.isin(_pack(mid.C.cpu().numpy()), _pack(_i16(case['mid_coords_i16'])))
        assert member.sum() == len(_i16(case['mid_coords_i16']))
        mid = ME.MinkowskiPruning()(mid, torch.from_numpy(member).cuda())    # key (2, 'pruned')
        cand = up(mid)
    want_key = _pack(_i16(case['cand_coords_i16']))
    got_key = _pack(cand.C.cpu().numpy())
    order = np.argsort(want_key)
    pos = order[np.searchsorted(want_key[order], got_key)]                   # fixture row of every engine row
    assert (want_key[pos] == got_key).all() and len(want_key) == len(got_key)
    logits = np.frombuffer(bytes.fromhex(case['logits_f32']), dtype='<f4')[pos]
    pred = ME.SparseTensor(torch.from_numpy(logits.copy()).cuda().view(-1, 1), coordinate_map_key=cand.coordinate_map_key,
                           coordinate_manager=cm)
    return cm, mid, pred, pos


def _want(query, pos):
    bits = np.unpackbits(np.frombuffer(bytes.fromhex(query['keep']), dtype=np.uint8))
    return torch.from_numpy(bits[pos].astype(bool))


def _keep_through_pooling(ME, cm, pred, targets):
    """the rule in the engine's own operators: a candidate stays if it is above its sample's threshold or is the maximum of
    its stride-2 cell; the threshold is the (n - target)-th smallest among the sample's non-maximum logits"""
    cell_key = ME.CoordinateMapKey([2, 2, 2], 'pruned')
    assert cell_key in cm.get_coordinate_map_keys([2, 2, 2])
    pool = ME.MinkowskiMaxPooling(2, 2, dimension=3)
    spread = ME.MinkowskiPoolingTranspose(2, 2, dimension=3)
    cell_max = spread(pool(pred, cell_key), pred.coordinate_map_key).F.view(-1)
    v = pred.F.view(-1)
    is_max = v == cell_max
    if targets is None:
        return (v > 0) | is_max
    keep = torch.zeros_like(is_max)
    for rows, target in zip(pred.decomposition_permutations, targets):
        sample = v[rows]
        below = sample[~is_max[rows]]
        thr = torch.sort(below).values[sample.numel() - target - 1]
        keep[rows] = (sample > thr) | is_max[rows]
    return keep


def test_key_naming_follows_the_reference_expectations():
    from fastpcc_amd import engine as ME
    cm, mid, pred, _ = _rebuild(ME, CASES['one_cloud_a'])
    keys = cm.get_coordinate_map_keys([2, 2, 2])
    assert sorted(k.get_key()[1] for k in keys) == ['', 'pruned']            # generated set '' + its pruning, as ME names them
    assert mid.coordinate_map_key == ME.CoordinateMapKey([2, 2, 2], 'pruned')
    assert pred.coordinate_map_key.get_key()[1] == ''


@pytest.mark.parametrize('label', ['one_cloud_a', 'one_cloud_b'])
def test_fused_kernel_and_pooling_operators_reproduce_the_reference_masks(label):
    from fastpcc_amd import engine as ME, hipops as ops
    case = CASES[label]
    cm, mid, pred, pos = _rebuild(ME, case)
    for q in case['queries']:
        want = _want(q, pos).cuda()
        assert int(want.sum()) == q['kept']
        assert torch.equal(_keep_through_pooling(ME, cm, pred, q['points_num']), want), q['points_num']
        if q['points_num'] is not None:
            assert torch.equal(ops.topk_keep(pred.F.view(-1), q['points_num'][0]).bool(), want), q['points_num']
        else:                                                                # adaptive_pruning = False: threshold 0
            cells = pred.F.view(-1, 8)
            assert torch.equal(((cells > 0) | (cells == cells.max(1, keepdim=True).values)).view(-1), want)


def test_per_sample_thresholds_of_a_batch():
    """two clouds in one batch: one k-th value per sample (decomposition_permutations), as the training-time pruning uses it"""
    from fastpcc_amd import engine as ME, hipops as ops
    case = CASES['two_clouds']
    cm, mid, pred, pos = _rebuild(ME, case)
    perms = pred.decomposition_permutations
    assert len(perms) == 2 and sum(p.numel() for p in perms) == pred.shape[0]
    for q in case['queries']:
        want = _want(q, pos).cuda()
        assert torch.equal(_keep_through_pooling(ME, cm, pred, q['points_num']), want)
        if q['points_num'] is not None:
            got = torch.empty_like(want)
            for p, t in zip(perms, q['points_num']):
                got[p] = ops.topk_keep(pred.F.view(-1)[p].contiguous(), t).bool()
            assert torch.equal(got, want)
    origin_key, rows = cm.origin_map(pred.coordinate_map_key)
    assert [r.tolist() for r in rows] == [p.tolist() for p in perms]


def test_general_kernel_maps_match_the_oracle():
    from fastpcc_amd import engine as ME
    from oracle import coords as oc
    xyz = surface_cloud(4, 64, 5000)
    coords = torch.from_numpy(batched(xyz)).to(torch.int32).cuda()
    cm = ME.CoordinateManager(D=3)
    x = ME.SparseTensor(torch.ones((len(xyz), 1), device='cuda'), coordinates=coords, coordinate_manager=cm)
    key = x.coordinate_map_key
    up = cm.stride(key, 2)
    lvl = oc.Level(batched(xyz), 1)
    for (a, b, ks, st, want) in ((key, key, 3, 1, oc.kernel_map(lvl, lvl, 3)), (key, up, 2, 2, oc.kernel_map(lvl, oc.strided(lvl), 2))):
        got = cm.kernel_map(a, b, stride=st, kernel_size=ks)
        assert sorted(got) == [k for k, (i, _) in enumerate(want) if len(i)]
        for k, pair in got.items():
            assert pair[0].cpu().tolist() == want[k][0].tolist() and pair[1].cpu().tolist() == want[k][1].tolist()


def test_batchnorm_wrapper_keys():
    from fastpcc_amd import engine as ME
    bn = ME.MinkowskiBatchNorm(8)
    assert sorted(bn.state_dict()) == ['bn.bias', 'bn.num_batches_tracked', 'bn.running_mean', 'bn.running_var', 'bn.weight']
