"""Dataset classes and the device-capable sample pipeline (fastpcc_amd/datasets.py, the kd-tree partition of
fastpcc_amd/data.py) on the host; the GPU twin of these checks is tests/test_gpu_datasets.py."""
import hashlib
import json
import os

import numpy as np
import pytest
import torch

from fastpcc_amd import datasets as D
from fastpcc_amd.data import _kd_tree_partition_device, write_ply_file

G = json.load(open(os.path.join(os.path.dirname(__file__), 'golden', 'kdtree.json')))


@pytest.mark.parametrize('case', G, ids=lambda c: f"n{c['n']}-max{c['max_num']}")
def test_tensor_kd_tree_partition_matches_reference(case):
    """the tensor-op form (what runs on the GPU) against the reference's own partitions"""
    rng = np.random.default_rng(case['seed'])
    coord = rng.integers(0, case['range'], (case['n'], 3)).astype(np.int32)
    parts, (ids,) = _kd_tree_partition_device(torch.from_numpy(coord), case['max_num'], [torch.arange(case['n'])])
    assert [len(p) for p in parts] == case['sizes']
    assert [hashlib.sha256(np.ascontiguousarray(p.numpy()).tobytes()).hexdigest()[:16] for p in parts] == case['sha']
    assert all((coord[i.numpy()] == p.numpy()).all() for i, p in zip(ids, parts))


def test_voxelize_equals_numpy_restatement():
    rng = np.random.default_rng(0)
    pts = (rng.normal(size=(5000, 3)) * 30).astype(np.float32)
    vox, org = D.voxelize(pts, 4095 / 400)
    want = pts - pts.min(0)
    want *= np.float32(4095 / 400)
    want = np.unique(want.round().astype(np.int32), axis=0)
    assert (vox.numpy() == want).all() and np.allclose(org.numpy(), pts.min(0))
    order = D.morton_order(vox)
    key = lambda v: sum(((v[:, a].astype(np.int64) >> b) & 1) << (3 * b + a) for b in range(21) for a in range(3))
    assert (np.diff(key(vox.numpy()[order.numpy()])) > 0).all()
    inv = D.morton_order(vox, inverse=True)
    assert (np.diff(key(vox.numpy()[inv.numpy()][:, ::-1])) > 0).all()


def test_random_partition_keeps_a_slab():
    g = torch.Generator().manual_seed(3)
    coord = torch.from_numpy(np.unique(np.random.default_rng(1).integers(0, 300, (20000, 3)).astype(np.int32), axis=0))
    ids = torch.arange(coord.shape[0])
    part, (pid,) = D.kd_tree_partition_randomly(coord, 3000, (ids,), generator=g)
    assert 0 < part.shape[0] <= 3000 * 1.1 and (coord[pid] == part).all()
    assert D.kd_tree_partition_randomly(coord[:100], 3000) is not None and D.kd_tree_partition_randomly(coord[:100], 3000).shape[0] == 100


def test_kitti_dataset(tmp_path):
    rng = np.random.default_rng(5)
    for seq, names in (('00', ('000000', '000001')), ('11', ('000000',))):
        os.makedirs(tmp_path / seq / 'velodyne')
        for n in names:
            np.concatenate(((rng.normal(size=(3000, 3)) * 20).astype('<f4'), np.zeros((3000, 1), '<f4')), 1).tofile(tmp_path / seq / 'velodyne' / f'{n}.bin')
    cfg = D.KITTIOdometryConfig(root=str(tmp_path), resolution=4096, morton_sort=True, kd_tree_partition_max_points_num=1000)
    test = D.KITTIOdometry(cfg, False)
    assert len(test) == 1 and os.path.exists(tmp_path / 'test_list.txt')
    s = test[0]
    assert s.xyz.dtype == torch.int32 and s.xyz.min() == 0 and s.org_points_num == 3000 and s.resolution == 59.70 + 1
    assert s.inv_transform.shape == (4,) and abs(float(s.inv_transform[3]) - 400 / 4095) < 1e-7
    batch = test.collate_fn([s])
    assert isinstance(batch.xyz, list) and sum(p.shape[0] for p in batch.xyz[1:]) == s.xyz.shape[0]
    train = D.KITTIOdometry(D.KITTIOdometryConfig(root=str(tmp_path), random_flip=True, random_rotation=True,
                                                  kd_tree_partition_max_points_num=800), True)
    assert len(train) == 2
    b = train.collate_fn([train[0], train[1]])
    assert b.batch_size == 2 and b.xyz.shape[1] == 4 and set(b.xyz[:, 0].tolist()) == {0, 1}


def test_shapenet_dataset_samples_meshes_and_caches(tmp_path):
    d = tmp_path / '0001' / 'abc' / 'models'
    os.makedirs(d)
    with open(d / 'model_normalized.obj', 'w') as f:                  # a unit cube, quads
        for v in [(x, y, z) for x in (0, 1) for y in (0, 1) for z in (0, 1)]:
            f.write('v %d %d %d\n' % v)
        for q in ((1, 2, 4, 3), (5, 7, 8, 6), (1, 5, 6, 2), (3, 4, 8, 7), (1, 3, 7, 5), (2, 6, 8, 4)):
            f.write('f %d %d %d %d\n' % q)
    pts = D.sample_mesh_uniform(str(d / 'model_normalized.obj'), 20000, np.random.default_rng(0))
    on_face = (np.isclose(pts, 0) | np.isclose(pts, 1)).any(1)
    assert on_face.all() and pts.min() >= -1e-12 and pts.max() <= 1 + 1e-12
    per_face = [int((np.isclose(pts[:, a], v)).sum()) for a in range(3) for v in (0, 1)]
    assert min(per_face) > 0.8 * 20000 / 6                                # area-uniform: equal faces get equal shares
    cfg = D.ShapeNetCorev2Config(root=str(tmp_path), mesh_sample_points_num=20000, mesh_sample_point_resolution=64, resolution=32,
                                 random_rotation=False)
    ds = D.ShapeNetCorev2(cfg, True)
    a = ds[0]
    assert a.xyz.dtype == torch.int32 and a.xyz.min() >= 0 and a.xyz.max() <= 32
    assert os.path.isfile(ds._cache_path(ds.file_list[0]))
    assert torch.equal(ds[0].xyz, a.xyz)                                  # second read comes from the cache: no new sampling
    assert len(torch.unique(a.xyz, dim=0)) == a.xyz.shape[0]


def test_ply_voxel_dataset_with_colour_and_partitions(tmp_path):
    rng = np.random.default_rng(2)
    xyz = np.unique(rng.integers(0, 200, (6000, 3)).astype(np.int32), axis=0)
    rgb = rng.integers(0, 256, xyz.shape).astype(np.uint8)
    os.makedirs(tmp_path / 'set')
    write_ply_file(xyz + 7, str(tmp_path / 'set' / 'frame_0001.ply'), rgb=rgb)
    cfg = D.PlyVoxelConfig(root=str(tmp_path / 'set'), resolution=256, with_color=True, morton_sort=True, kd_tree_partition_max_points_num=2500)
    ds = D.PlyVoxel(cfg, False)
    assert len(ds) == 1 and ds[0] == 0
    batch = ds.collate_fn([0])
    assert isinstance(batch.xyz, list) and batch.xyz[0].shape[0] == len(xyz) and len(batch.color) == len(batch.xyz)
    whole = batch.xyz[0][:, 1:].numpy()
    lut = {tuple(p): tuple(c) for p, c in zip(xyz.tolist(), rgb.tolist())}
    assert all(lut[tuple(p)] == tuple(int(v) for v in c) for p, c in zip(whole[:200].tolist(), batch.color[0][:200].tolist()))   # colours follow their voxels
    assert batch.inv_transform[0][:3].tolist() == [7.0, 7.0, 7.0] and batch.resolution == [256]
    assert sum(p.shape[0] for p in batch.xyz[1:]) == len(xyz) and max(p.shape[0] for p in batch.xyz[1:]) <= 2500
    rescaled = D.PlyVoxel(D.PlyVoxelConfig(root=str(tmp_path / 'set'), resolution=128, coord_scaler=0.5), True)
    b = rescaled.collate_fn([0, 0])
    assert b.batch_size == 2 and b.xyz[:, 1:].max() <= 100
