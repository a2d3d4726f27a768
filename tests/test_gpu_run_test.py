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
