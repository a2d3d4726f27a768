"""lossl_coord_int on the GPU against the integer oracle: the BITSTREAM must be byte-identical (BASELINE.json:
'bit-exact bitstream for lossl_coord_int'), decoding must be lossless and the two implementations must decode each other's
streams."""
import numpy as np
import pytest
import torch

from fastpcc_amd.synthetic import batched, lidar_cloud
from oracle.codec_int import OracleInt

pytestmark = pytest.mark.gpu


def _model(channels, skip, seed):
    from fastpcc_amd.codecs.lossl_coord_int import Config, Model
    from fastpcc_amd.codecs.lossl_coord_int.init_random import randomize_
    cfg = Config(channels=channels, skip_top_scales_num=skip)
    model = Model(cfg, 'cuda')
    randomize_(model, seed)
    weights = {k: v.clone() for k, v in model.state_dict().items()}
    return cfg, model.cuda().eval(), weights


@pytest.mark.parametrize('channels,skip,beams,az', [(32, 0, 16, 512), (64, 1, 24, 512), (256, 0, 16, 256)])
def test_bitstream_identical_and_lossless(channels, skip, beams, az):
    cfg, model, weights = _model(channels, skip, 5)
    xyz = lidar_cloud(3, beams=beams, azimuths=az) + np.array([11, 0, 5], dtype=np.int32)
    coords = batched(xyz)
    dev = torch.from_numpy(coords).cuda()
    perm = torch.randperm(len(xyz), generator=torch.Generator().manual_seed(0)).cuda()
    data = model.compress(dev[perm])
    oracle = OracleInt(weights, cfg)
    want = oracle.compress(coords.astype(np.int64))
    assert data == want
    rec = model.decompress(data).cpu().numpy()
    assert sorted(map(tuple, rec.tolist())) == sorted(map(tuple, xyz.tolist()))
    assert (oracle.decompress(data) == rec).all()                    # same points in the same (Morton) order
    assert model.compress(dev) == data


def test_state_dict_roundtrip_keeps_uint32_multipliers():
    cfg, model, weights = _model(32, 0, 9)
    assert weights['block_dec_recurrent.dec.conv_prelu.requant_mul'].dtype == torch.int64       # saved as int64
    from fastpcc_amd.codecs.lossl_coord_int import Model
    other = Model(cfg, 'cuda').cuda().eval()
    other.load_state_dict(weights)
    assert other.block_dec_recurrent.dec.conv_prelu.requant_mul.dtype == torch.uint32
    xyz = lidar_cloud(4, beams=8, azimuths=256)
    dev = torch.from_numpy(batched(xyz)).cuda()
    assert other.compress(dev) == model.compress(dev)


def test_test_forward_and_partitions():
    from fastpcc_amd.data import PCData
    cfg, model, _ = _model(32, 0, 2)
    xyz = lidar_cloud(5, beams=16, azimuths=512)
    dev = torch.from_numpy(batched(xyz)).cuda()
    out = model(PCData(xyz=dev))
    assert out['pred'].shape == (len(xyz), 3) and out['bpp'] > 0
    halves = dev[dev[:, 1] < int(xyz[:, 0].mean())].contiguous(), dev[dev[:, 1] >= int(xyz[:, 0].mean())].contiguous()
    rec = model.decompress_partitions(model.compress_partitions([dev, *halves])).cpu().numpy()
    assert sorted(map(tuple, rec.tolist())) == sorted(map(tuple, xyz.tolist()))


def _golden_runs():
    import json, os
    with open(os.path.join(os.path.dirname(__file__), 'golden', 'codec_int.json')) as f:
        return json.load(f)['runs']


@pytest.mark.parametrize('run', _golden_runs(), ids=[r['label'] for r in _golden_runs()])
def test_stream_identical_to_the_reference_run(run):
    """tests/golden/codec_int.json holds streams written by the reference's own model code and coder (make_golden.py): the
    HIP path must write the same bytes and decode them to the same points"""
    import hashlib
    from fastpcc_amd.codecs.lossl_coord_int import Config, Model
    from fastpcc_amd.codecs.lossl_coord_int.init_random import randomize_
    model = Model(Config(**run['config']), 'cuda')
    randomize_(model, run['seed'])
    model = model.cuda().eval()
    xyz = np.array(run['xyz'], dtype=np.int32)
    dev = torch.from_numpy(batched(xyz)).cuda()
    want = bytes.fromhex(run['stream_hex'])
    assert model.compress(dev) == want
    rec = model.decompress(want).cpu().numpy().astype(np.int32)
    assert hashlib.sha256(np.ascontiguousarray(rec).tobytes()).hexdigest() == run['recon_sha256']
