"""The MinkowskiEngine names the reference's decoder reaches for directly -- MinkowskiMaxPooling, MinkowskiPoolingTranspose,
CoordinateMapKey look-ups by string id, decomposition_permutations, kernel_map, origin_map -- on fastpcc_amd.engine.

`reference_get_keep` below is the CALL SEQUENCE of Decoder.get_keep
(/root/reference/models/convolutional/lossy_coord_v2/layers.py:151-180) written against a module handle `ME`; it is run on
the engine and must agree bit for bit with the fused product path (fpcc_topk_keep), which never builds those tensors."""
import numpy as np
import pytest
import torch

from util import batched, surface_cloud

pytestmark = pytest.mark.gpu


def reference_get_keep(ME, pred, points_num_list, max_stride_lossy_recon, training=False):
    _cm = pred.coordinate_manager._manager
    max_stride_coord_key = ME.CoordinateMapKey(
        max_stride_lossy_recon, '' if training or len(_cm.get_coordinate_map_keys(max_stride_lossy_recon)) == 1 else 'pruned')
    stride_scaler = [a // b for a, b in zip(max_stride_coord_key.get_tensor_stride(), pred.tensor_stride)]
    pool = ME.MinkowskiMaxPooling(stride_scaler, stride_scaler, dimension=3).to(pred.device)
    un_pool = ME.MinkowskiPoolingTranspose(stride_scaler, stride_scaler, dimension=3).to(pred.device)
    pred_local_max = un_pool(pool(pred, max_stride_coord_key), pred.coordinate_map_key)
    local_max_mask = (pred.F - pred_local_max.F).squeeze(1) != 0
    if points_num_list is not None:
        target_points_num = points_num_list.pop()
        sample_threshold = []
        for sample_tgt, sample_permutation in zip(target_points_num, pred.decomposition_permutations):
            sample = pred.F[sample_permutation]
            assert sample.shape[0] > sample_tgt
            sample_masked = sample[local_max_mask[sample_permutation]]
            sample_threshold.append(torch.kthvalue(sample_masked, sample.shape[0] - sample_tgt, dim=0).values)
        threshold = torch.tensor(sample_threshold, device=pred.F.device, dtype=pred.F.dtype)
        threshold = threshold[pred.C[:, 0].to(torch.long)]
    else:
        threshold = 0
    keep = (pred.F.squeeze(dim=1) > threshold)
    keep.logical_or_(~local_max_mask)
    return keep


def _decoder_like_candidates(ME, seed, batch=1):
    """what the decoder holds when it calls get_keep: logits on the generated children of a PRUNED stride-2 map"""
    clouds = [np.concatenate((np.full((len(x), 1), b), x), 1) for b, x in
              enumerate(np.unique(surface_cloud(seed + b, 64, 6000) // 4 * 4, axis=0) for b in range(batch))]
    coords = torch.from_numpy(np.concatenate(clouds)).to(torch.int32).cuda()
    cm = ME.CoordinateManager(D=3)
    top = ME.SparseTensor(torch.ones((coords.shape[0], 1), device='cuda'), coordinates=coords, tensor_stride=4, coordinate_manager=cm)
    up = ME.MinkowskiGenerativeConvolutionTranspose(1, 1, 2, 2, bias=False, dimension=3).cuda()
    with torch.no_grad():
        g = torch.Generator(device='cuda').manual_seed(seed)
        mid = up(top)                                                         # all 8 children at stride 2: key (2, '')
        mask = torch.rand(mid.shape[0], generator=g, device='cuda') < 0.4
        mask[::8] = True                                                      # every parent keeps a child
        mid = ME.MinkowskiPruning()(mid, mask)                                # key (2, 'pruned')
        cand = up(mid)                                                        # candidates at stride 1
        logits = torch.randn((cand.shape[0], 1), generator=g, device='cuda')
    pred = ME.SparseTensor(logits, coordinate_map_key=cand.coordinate_map_key, coordinate_manager=cm)
    return cm, mid, pred


def test_key_naming_follows_the_reference_expectations():
    from fastpcc_amd import engine as ME
    cm, mid, pred = _decoder_like_candidates(ME, 3)
    keys = cm.get_coordinate_map_keys([2, 2, 2])
    assert sorted(k.get_key()[1] for k in keys) == ['', 'pruned']            # generated set '' + its pruning, as ME names them
    assert mid.coordinate_map_key == ME.CoordinateMapKey([2, 2, 2], 'pruned')
    assert pred.coordinate_map_key.get_key()[1] == ''


@pytest.mark.parametrize('seed', [1, 2])
def test_reference_get_keep_sequence_equals_the_fused_kernel(seed):
    from fastpcc_amd import engine as ME, hipops as ops
    cm, mid, pred = _decoder_like_candidates(ME, seed)
    n = pred.shape[0]
    for target in (mid.shape[0], n // 3, n - 9):
        want = reference_get_keep(ME, pred, [[target]], [2, 2, 2])
        got = ops.topk_keep(pred.F.view(-1), target).bool()
        assert torch.equal(got, want), target
    # adaptive_pruning = False: threshold 0
    want = reference_get_keep(ME, pred, None, [2, 2, 2])
    cells = pred.F.view(-1, 8)
    assert torch.equal(((cells > 0) | (cells == cells.max(1, keepdim=True).values)).view(-1), want)


def test_reference_get_keep_sequence_per_sample_thresholds():
    """two clouds in one batch: one k-th value per sample (decomposition_permutations), as the training-time pruning uses it"""
    from fastpcc_amd import engine as ME, hipops as ops
    cm, mid, pred = _decoder_like_candidates(ME, 7, batch=2)
    perms = pred.decomposition_permutations
    assert len(perms) == 2 and sum(p.numel() for p in perms) == pred.shape[0]
    targets = [perms[0].numel() // 4, perms[1].numel() // 2]
    want = reference_get_keep(ME, pred, [list(targets)], [2, 2, 2])
    got = torch.cat([ops.topk_keep(pred.F.view(-1)[p], t).bool() for p, t in zip(perms, targets)])
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
