"""The sample pipeline of fastpcc_amd/datasets.py and the kd-tree partition with the cloud resident on the GPU: the same
partitions as the reference's (tests/golden/kdtree.json), the same voxels and Morton order as on the host, and a frame that
leaves the dataset in HBM goes straight into the codec."""
import hashlib
import json
import os

import numpy as np
import pytest
import torch

from fastpcc_amd import datasets as D
from fastpcc_amd.data import PCData, kd_tree_partition, pc_data_collate_fn, write_ply_file

pytestmark = pytest.mark.gpu
G = json.load(open(os.path.join(os.path.dirname(__file__), 'golden', 'kdtree.json')))


@pytest.mark.parametrize('case', G, ids=lambda c: f"n{c['n']}-max{c['max_num']}")
def test_kd_tree_partition_on_the_device_matches_reference(case):
    rng = np.random.default_rng(case['seed'])
    coord = rng.integers(0, case['range'], (case['n'], 3)).astype(np.int32)
    parts, (ids,) = kd_tree_partition(torch.from_numpy(coord).cuda(), case['max_num'], [torch.arange(case['n']).cuda()])
    assert all(p.is_cuda for p in parts)
    assert [len(p) for p in parts] == case['sizes']
    assert [hashlib.sha256(np.ascontiguousarray(p.cpu().numpy()).tobytes()).hexdigest()[:16] for p in parts] == case['sha']


def test_voxelize_and_morton_order_on_the_device_equal_the_host():
    rng = np.random.default_rng(1)
    pts = (rng.normal(size=(200000, 3)) * 40).astype(np.float32)
    host, org_h = D.voxelize(pts, 4095 / 400)
    dev, org_d = D.voxelize(pts, 4095 / 400, device='cuda')
    assert dev.is_cuda and torch.equal(dev.cpu(), host) and torch.equal(org_d.cpu(), org_h)
    for inverse in (False, True):
        assert torch.equal(dev[D.morton_order(dev, inverse)].cpu(), host[D.morton_order(host, inverse)])


def test_ply_frame_from_the_dataset_goes_straight_into_the_codec(tmp_path):
    from fastpcc_amd.codecs.lossy_coord_v2 import Model
    from fastpcc_amd.codecs.lossy_coord_v2.model_config import baseline_r1
    from fastpcc_amd.synthetic import enliven, surface_cloud
    xyz = surface_cloud(4, 128, 50000)
    os.makedirs(tmp_path / 'set')
    write_ply_file(xyz + 3, str(tmp_path / 'set' / 'frame_0001.ply'))
    ds = D.PlyVoxel(D.PlyVoxelConfig(root=str(tmp_path / 'set'), resolution=128, kd_tree_partition_max_points_num=20000), False,
                    device='cuda')
    batch = ds.collate_fn([0])
    assert isinstance(batch.xyz, list) and all(p.is_cuda for p in batch.xyz)          # whole frame + partitions, all in HBM
    assert sum(p.shape[0] for p in batch.xyz[1:]) == len(xyz)
    torch.manual_seed(0)
    model = Model(baseline_r1())
    enliven(model, 0)
    model = model.cuda().eval()
    out = model(batch)
    assert out['pred'].shape[0] == len(xyz) and out['bpp'] > 0
