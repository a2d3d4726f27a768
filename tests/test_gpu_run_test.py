"""fastpcc_amd.run_test -- the evaluation driver in the role of the reference's test.py: a YAML file of the reference's format,
PLY input, kd-tree partitioned coding, per-file and mean metrics."""
import json

import numpy as np
import pytest
import torch

from util import surface_cloud

pytestmark = pytest.mark.gpu

YAML = """\
model_module_path: models.convolutional.lossy_coord_v2
model:
  activation: 'prelu'
  compressed_channels: [1]
  skip_encoding_fea: 1
  encoder_channels: [16, 64]
  decoder_channels: [16]
  adaptive_pruning: True
  geo_lossl_if_sample: [0, 1, 0, 1, 0, 1, 0, 1, 0, 1, 0, 1]
  geo_lossl_channels: [64, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 1]
  bits_loss_factor: 0.4
test:
  dataset:
    kd_tree_partition_max_points_num: [9000, 0, 0]
"""


def test_driver_on_ply_files(tmp_path, capsys):
    from fastpcc_amd.data import write_ply_file
    from fastpcc_amd.run_test import main
    cfg = tmp_path / 'cfg.yaml'
    cfg.write_text(YAML)
    small, big = surface_cloud(51, 64, 6000), surface_cloud(52, 128, 30000)
    write_ply_file(small.astype(np.float32), str(tmp_path / 'small.ply'))
    write_ply_file(big.astype(np.float32), str(tmp_path / 'big.ply'), rgb=np.zeros_like(big, dtype=np.uint8))
    torch.manual_seed(0)
    assert main(['--config', str(cfg), '--ply', str(tmp_path / 'small.ply'), str(tmp_path / 'big.ply'),
                 '--results-dir', str(tmp_path / 'out')]) == 0
    report = json.loads(capsys.readouterr().out)
    assert set(report['files']) == {'small.ply', 'big.ply'}
    assert report['mean']['samples_num'] == 2 and report['mean']['bpp(mean)'] > 0
    assert 'mseF,PSNR (p2point)(mean)' in report['mean']
    assert (tmp_path / 'out' / 'bin' / 'small.bin').stat().st_size > 0          # bitstreams written like the reference does
    assert (tmp_path / 'out' / 'bin' / 'big.bin').stat().st_size > 0
    assert (tmp_path / 'out' / 'mean_metric.json').exists()
    # the big cloud (> 9000 points) went through kd-tree partitions: its stream is a sequence of length-prefixed parts
    data = (tmp_path / 'out' / 'bin' / 'big.bin').read_bytes()
    at, parts = 0, 0
    while at < len(data):
        at += 3 + int.from_bytes(data[at:at + 3], 'little')
        parts += 1
    assert at == len(data) and parts >= 2


def test_driver_on_a_kitti_sweep_with_the_integer_codec(tmp_path, capsys):
    """velodyne .bin -> voxelisation of the reference dataset -> lossl_coord_int -> points back in metres: every decoded point is
    the centre of the voxel of some input point (error <= half a cell per axis)"""
    from fastpcc_amd.codecs.lossl_coord_int import Config, Model
    from fastpcc_amd.codecs.lossl_coord_int.init_random import randomize_
    from fastpcc_amd.data import kitti_odometry_sample, pc_data_collate_fn
    rng = np.random.default_rng(3)
    n = 20000
    r, az = rng.uniform(3, 70, n), rng.uniform(0, 2 * np.pi, n)
    pts = np.stack([r * np.cos(az), r * np.sin(az), rng.normal(-1.6, 0.3, n), rng.uniform(0, 1, n)], 1).astype('<f4')
    path = tmp_path / '000001.bin'
    pts.tofile(path)
    model = Model(Config(channels=32), 'cuda')
    randomize_(model, 4)
    model = model.cuda().eval()
    sample = kitti_odometry_sample(str(path), 4096, device='cuda')
    batch = pc_data_collate_fn([PC for PC in [sample]]).to('cuda')
    out = model(batch)
    rec = out['pred'].cpu().numpy()                                   # metres, via inv_transform
    assert rec.shape == (sample.xyz.shape[0], 3)
    inv = sample.inv_transform[0].numpy()
    want = sample.xyz.cpu().numpy().astype(np.float32) * inv[3] + inv[:3]
    assert np.allclose(np.sort(rec.view([('', rec.dtype)] * 3), axis=0).view(rec.dtype).reshape(-1, 3),
                       np.sort(want.view([('', want.dtype)] * 3), axis=0).view(want.dtype).reshape(-1, 3), atol=1e-4)
    # the first 16 bytes of the stream carry the transform (model.py: inv_transform prepended for the evaluator)
    assert out['compressed_bytes'][:16] == sample.inv_transform[0].numpy().astype('<f4').tobytes()
    assert out['bpp'] > 0
