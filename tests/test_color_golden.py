"""lossy_coord_lossy_color against tests/golden/codec_color.json: the REFERENCE's own model code and coders executed on the
CPU over the functional MinkowskiEngine stand-in of tests/golden/make_golden.py (see tests/test_v2_golden.py for what that
pins and what stays restated)."""
import json
import os
from dataclasses import fields

import numpy as np
import pytest
import torch

from fastpcc_amd.codecs.lossy_coord_lossy_color import Model
from fastpcc_amd.codecs.lossy_coord_lossy_color.model_config import ModelConfig
from fastpcc_amd.synthetic import enliven
from oracle.codec_color import OracleColor
from test_v2_golden import same_float_behaviour

with open(os.path.join(os.path.dirname(__file__), 'golden', 'codec_color.json')) as f:
    RUNS = json.load(f)['runs']


def model_of(run):
    known = {f.name for f in fields(ModelConfig)}
    assert set(run['config']) <= known, set(run['config']) - known          # the reference's configuration fields exist here
    cfg = ModelConfig(**{k: tuple(v) if isinstance(v, list) else v for k, v in run['config'].items()})
    torch.manual_seed(0)
    model = Model(cfg)
    enliven(model, run['seed'])
    return cfg, model


@pytest.mark.parametrize('run', RUNS, ids=[r['label'] for r in RUNS])
def test_reference_run(run):
    cfg, model = model_of(run)
    assert float(sum(p.detach().double().abs().sum() for n, p in model.named_parameters() if '.prior_' not in n)) == \
        pytest.approx(run['param_abs_sum'], rel=1e-12)
    xyz = np.array(run['xyz'], dtype=np.int64)
    coords = np.concatenate((np.zeros((len(xyz), 1), np.int64), xyz), 1)
    color = np.array(run['color'], dtype=np.uint8)
    want = bytes.fromhex(run['stream_hex'])
    weights = {k: v for k, v in model.state_dict().items() if isinstance(v, torch.Tensor)}
    oracle = OracleColor(weights, cfg, conv='mm')
    data = oracle.compress(coords, color)
    head = 6 + 3 * (len(cfg.encoder_channels) - 1)
    assert data[:head + 4] == want[:head + 4]
    if same_float_behaviour():
        assert data == want
        rec_xyz, rec_rgb = oracle.decompress(want)
        assert rec_xyz.tolist() == run['recon_xyz']
        assert np.asarray(rec_rgb).astype(int).tolist() == run['recon_rgb']
    else:
        assert abs(len(data) - len(want)) <= 0.02 * len(want) + 4


with open(os.path.join(os.path.dirname(__file__), 'golden', 'codec_color_chain.json')) as f:
    GC = json.load(f)


def chain_model_of(run):
    known = {f.name for f in fields(ModelConfig)}
    cfg = ModelConfig(**{k: tuple(v) if isinstance(v, list) else v for k, v in run['config'].items() if k in known})
    torch.manual_seed(0)
    model = Model(cfg)
    enliven(model, run['seed'], **({} if run.get('gain') is None else {'gain': run['gain']}))
    return cfg, model


@pytest.mark.parametrize('run', GC['runs'], ids=[r['label'] for r in GC['runs']])
def test_reference_run_in_chain_order(run):
    """codec_color_chain.json: the reference's colour codec run with every layer summed in the order the HIP kernels document
    (make_golden.py:make_codec_color_chain); the chain-mode oracle writes the same bytes and decodes the same coloured cloud"""
    from fastpcc_amd import hipops
    from fastpcc_amd.engine import summation_order
    assert GC['numerics_version'] == hipops.numerics_version(), 'numerics version bumped: regenerate codec_color_chain.json'
    cfg, model = chain_model_of(run)
    assert float(sum(p.detach().double().abs().sum() for n, p in model.named_parameters() if '.prior_' not in n)) == \
        pytest.approx(run['param_abs_sum'], rel=1e-12)
    xyz = np.array(run['xyz'], dtype=np.int64)
    coords = np.concatenate((np.zeros((len(xyz), 1), np.int64), xyz), 1)
    want = bytes.fromhex(run['stream_hex'])
    weights = {k: v for k, v in model.state_dict().items() if isinstance(v, torch.Tensor)}
    oracle = OracleColor(weights, cfg, conv='chain', order_fn=summation_order)
    assert oracle.compress(coords, np.array(run['color'], dtype=np.uint8)) == want
    rec_xyz, rec_rgb = oracle.decompress(want)
    assert rec_xyz.tolist() == run['recon_xyz']
    assert np.asarray(rec_rgb).astype(int).tolist() == run['recon_rgb']


# the reference's LIST path of the colour codec in chain order (codec_color_partitions_chain.json, round 5; see test_v2_golden.py)
with open(os.path.join(os.path.dirname(__file__), 'golden', 'codec_color_partitions_chain.json')) as f:
    GP = json.load(f)


@pytest.mark.parametrize('run', GP['runs'], ids=[r['label'] for r in GP['runs']])
def test_reference_partition_lists_in_chain_order(run):
    from oracle.orders import summation_order, NUMERICS_VERSION
    assert GP['numerics_version'] == NUMERICS_VERSION, 'numerics version bumped: regenerate codec_color_partitions_chain.json'
    cfg, model = chain_model_of(run)
    weights = {k: v for k, v in model.state_dict().items() if isinstance(v, torch.Tensor)}
    oracle = OracleColor(weights, cfg, conv='chain', order_fn=summation_order)
    blob = bytes.fromhex(run['blob_hex'])
    pos = 0
    for xyz, color, part in zip(run['xyz'], run['color'], run['parts']):
        length = int.from_bytes(blob[pos:pos + 3], 'little')
        stream = blob[pos + 3: pos + 3 + length]
        xyz = np.array(xyz, dtype=np.int64)
        assert oracle.compress(np.concatenate((np.zeros((len(xyz), 1), np.int64), xyz), 1), np.array(color, dtype=np.uint8)) == stream
        rec_xyz, rec_rgb = oracle.decompress(stream)
        assert rec_xyz.tolist() == part['recon_xyz'] and np.asarray(rec_rgb).astype(int).tolist() == part['recon_rgb']
        pos += 3 + length
    assert pos == len(blob)
