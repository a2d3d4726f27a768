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


def test_several_clouds_in_one_traversal_write_the_single_cloud_streams():
    """compress_many / decompress_many of the integer codec: B LiDAR sweeps as the samples of one batch, one launch per operator and
    level over all of them, every cloud's stream byte-identical to the one it gets alone (and therefore to the oracle's) and decoded to
    the same points in the same order; partition lists go through the same path"""
    cfg, model, weights = _model(32, 0, 5)
    clouds = [lidar_cloud(3, beams=16, azimuths=512) + np.array([11, 0, 5], dtype=np.int32),
              lidar_cloud(4, beams=8, azimuths=256),
              lidar_cloud(6, beams=24, azimuths=384) + np.array([0, 7, 0], dtype=np.int32)]
    devs = [torch.from_numpy(batched(c)).cuda() for c in clouds]
    alone = [model.compress(d) for d in devs]
    recs = [model.decompress(s) for s in alone]
    oracle = OracleInt(weights, cfg)
    assert alone[1] == oracle.compress(batched(clouds[1]).astype(np.int64))
    for pick in ([0, 1, 2], [2, 0], [1, 1]):
        many = model.compress_many([devs[i] for i in pick])
        for i, s in zip(pick, many):
            assert s == alone[i], f'cloud {i} of batch {pick}: stream differs from the one coded alone'
        for i, r in zip(pick, model.decompress_many(many)):
            assert torch.equal(r, recs[i]), f'cloud {i} of batch {pick}: decoded points differ'
    blob = model.compress_partitions([torch.cat(devs), *devs])
    assert blob == b''.join(len(s).to_bytes(3, 'little') + s for s in alone)
    rec = model.decompress_partitions(blob)
    assert torch.equal(rec, torch.cat(recs))


def test_batched_streams_equal_the_reference_runs():
    """two clouds of the reference runs of codec_int.json that share a configuration, coded in one batch: each stream is the reference
    run's bytes"""
    from fastpcc_amd.codecs.lossl_coord_int import Config, Model
    from fastpcc_amd.codecs.lossl_coord_int.init_random import randomize_
    runs = _golden_runs()
    by_cfg = {}
    for r in runs:
        by_cfg.setdefault((str(sorted(r['config'].items())), r['seed']), []).append(r)
    checked = 0
    for group in by_cfg.values():
        run = group[0]
        model = Model(Config(**run['config']), 'cuda')
        randomize_(model, run['seed'])
        model = model.cuda().eval()
        mine = [torch.from_numpy(batched(np.array(r['xyz'], dtype=np.int32))).cuda() for r in group]
        other = torch.from_numpy(batched(lidar_cloud(9, beams=8, azimuths=256))).cuda()
        streams = model.compress_many([other, *mine])
        for r, s in zip(group, streams[1:]):
            assert s == bytes.fromhex(r['stream_hex'])
            checked += 1
        back = model.decompress_many(streams)
        for r, rec in zip(group, back[1:]):
            import hashlib
            assert hashlib.sha256(np.ascontiguousarray(rec.cpu().numpy().astype(np.int32)).tobytes()).hexdigest() == r['recon_sha256']
    assert checked == len(runs)


@pytest.mark.parametrize('clouds', [1, 3])
def test_octree_analysis_equals_get_bin(clouds):
    """Model.analyse (one counting pass + fpcc_octree_level per level, from the sorted keys) against the operator-by-operator form
    it replaces (get_bin: unique_consecutive + the fold convolution over a hash-table kernel map, the reference's own sequence):
    coordinates, occupancy bits, coded symbols and the (2, 2, 2) kernel map of every level, for one cloud and for a batch"""
    from fastpcc_amd import hipops as ops
    from fastpcc_amd.codecs.lossl_coord_int.model import _symbols_of
    cfg, model, _ = _model(32, 0, 5)
    parts = []
    for b in range(clouds):
        c = batched(lidar_cloud(3 + b, beams=8 + 4 * b, azimuths=256))
        c[:, 0] = b
        parts.append(c)
    xyz = torch.from_numpy(np.concatenate(parts)).cuda()
    keys = ops.morton3d_encode(xyz[:, 1:], (2, 1, 0)) | (xyz[:, 0].to(torch.int64) << 48)
    keys, perm = ops.sort_keys(keys)
    xyz = xyz[perm.long()].contiguous()
    levels = model.max_downsample_times
    mine, rows = model.analyse(xyz, keys, levels, clouds)
    ref = [model.get_init_pc(xyz, 1)]
    for _ in range(levels):
        ref.append(model.get_bin(ref[-1], ref[0].F))
    assert len(mine) == len(ref) == levels + 1
    for l in range(1, levels + 1):
        a, b = mine[l], ref[l]
        assert a.stride == b.stride and torch.equal(a.C, b.C), f'level {l}: coordinates'
        assert a.F.dtype == b.F.dtype and torch.equal(a.F, b.F), f'level {l}: occupancy bits'
        sym = a.F._fpcc_symbols
        plain = b.F.clone()                                       # no symbols attached: the tensor-operator form
        assert torch.equal(sym, _symbols_of(plain, model.bin2oct_kernel)), f'level {l}: symbols'
        assert rows[l] == torch.bincount(a.C[:, 0], minlength=clouds).tolist(), f'level {l}: rows per cloud'
        if l >= 2:
            tag = (ref[l - 1].stride, (2, 2, 2), (2, 2, 2))
            assert torch.equal(mine[0]._caches.kmaps[tag]['in_out_maps'], ref[0]._caches.kmaps[tag]['in_out_maps']), f'level {l}: kernel map'
            assert torch.equal(mine[0]._caches.cmaps[a.stride][0], ref[0]._caches.cmaps[b.stride][0])


def test_level_per_call_path_equals_the_layer_by_layer_path(monkeypatch):
    """fpcc_int_level_trunk / fpcc_int_level_expand (a OneScalePredictor level in two calls over a descriptor table of its layers)
    against the module-by-module traversal: same stream, same decoded points, and the intermediate tensors of one level bit for bit"""
    from fastpcc_amd.codecs.lossl_coord_int import model as M
    cfg, model, weights = _model(32, 0, 5)
    xyz = torch.from_numpy(batched(lidar_cloud(5, beams=16, azimuths=512))).cuda()
    one_scale = [b for b in [*model.blocks_dec, model.block_dec_recurrent] if isinstance(b, M.OneScalePredictor)]
    assert M.FAST_LEVELS and one_scale and all(b._described() is not None for b in one_scale)
    fast = model.compress(xyz)
    rec_fast = model.decompress(fast)
    monkeypatch.setattr(M, 'FAST_LEVELS', False)
    assert all(b._described() is None for b in one_scale)
    slow = model.compress(xyz)
    assert fast == slow
    assert torch.equal(model.decompress(slow), rec_fast)
    mixed = model.decompress(fast)                                   # layer-by-layer decoder on the level-per-call encoder's stream
    assert torch.equal(mixed, rec_fast)
    # one level in isolation: trunk + expand against _trunk + _expand on the same inputs
    block = model.block_dec_recurrent
    g = torch.Generator().manual_seed(3)
    coords = torch.unique(torch.cat((torch.zeros(300, 1, dtype=torch.int32), torch.randint(0, 12, (300, 3), generator=g, dtype=torch.int32)), 1), dim=0).cuda()
    n, c = coords.shape[0], cfg.channels
    feat = torch.randint(-(1 << 24), 1 << 24, (n, c), generator=g, dtype=torch.int32).cuda()
    symbols = torch.randint(0, 255, (n,), generator=g, dtype=torch.int16).cuda()
    count = int(M._children_count(symbols.cpu().numpy()))
    outs = []
    for flag in (True, False):
        monkeypatch.setattr(M, 'FAST_LEVELS', flag)
        x = M.SparseTensor(feat.clone(), coords, (4, 4, 4))
        d = block._described()
        assert (d is not None) == flag
        if flag:
            y, logits = block._trunk_fast(d, x)
            z = block._expand_fast(d, y, symbols, count, coords, None, block)
        else:
            y, logits = block._trunk(x)
            occ = M.Occupancy(symbols=symbols, count=count)
            z = block._expand(y, occ, occ.children(coords), _also=block._first_requant_of(block))
        q = z.F._fpcc_q8[id(block.dec.input_requant)]
        outs.append((y.F, logits, z.F, z.C, q))
    for a, b in zip(*outs):
        assert a.dtype == b.dtype and torch.equal(a, b)


def test_repeated_voxels_are_refused_not_miscoded():
    """the octree analysis counts its levels from the sorted keys and needs them unique (as the reference's data sets deliver them):
    a repeated voxel is an error from the counting pass, not a wrong stream"""
    cfg, model, _ = _model(32, 0, 5)
    xyz = batched(lidar_cloud(4, beams=8, azimuths=256))
    twice = torch.from_numpy(np.concatenate((xyz, xyz[100:103]))).cuda()
    with pytest.raises(ValueError, match='not unique'):
        model.compress(twice)
    assert len(model.compress(torch.from_numpy(xyz).cuda())) > 0          # the model is usable afterwards


@pytest.mark.parametrize('points', [
    [(0, 0, 0)],                                                    # one voxel: every level has one row
    [(5, 9, 2), (65535, 65535, 65535)],                             # two voxels a full 16-bit cube apart
    [(100, 100, 100), (100, 100, 101), (100, 101, 100), (101, 100, 100), (3000, 7, 9)],      # siblings and a stray
])
def test_tiny_clouds_equal_the_oracle(points):
    """the smallest maps: one row per level, a parent with several children beside a single voxel far away, coordinates at the 16-bit
    maximum -- stream bytes equal the oracle's, decoding is lossless, and the same clouds come through a batch unchanged"""
    cfg, model, weights = _model(32, 0, 5)
    xyz = np.array(points, dtype=np.int32)
    dev = torch.from_numpy(batched(xyz)).cuda()
    data = model.compress(dev)
    assert data == OracleInt(weights, cfg).compress(batched(xyz).astype(np.int64))
    rec = model.decompress(data).cpu().numpy()
    assert sorted(map(tuple, rec.tolist())) == sorted(map(tuple, xyz.tolist()))
    other = torch.from_numpy(batched(lidar_cloud(4, beams=8, azimuths=256))).cuda()
    many = model.compress_many([other, dev, dev])
    assert many[1] == data and many[2] == data and many[0] == model.compress(other)
    back = model.decompress_many(many)
    assert torch.equal(back[1], model.decompress(data)) and torch.equal(back[2], back[1])


def test_two_sweeps_in_flight_write_the_single_sweep_streams():
    """serving.FramePipeline over the integer codec: the second context is a copy of the modules over the same buffers (its layer
    descriptor tables are rebuilt, not copied: they hold raw pointers); streams and decoded points as alone"""
    from fastpcc_amd.serving import FramePipeline
    cfg, model, _ = _model(32, 0, 5)
    sweeps = [torch.from_numpy(batched(lidar_cloud(3 + i, beams=16, azimuths=384))).cuda() for i in range(4)]
    alone = [model.compress(s) for s in sweeps]                       # (also builds the descriptor tables that the copy must not share)
    recs = [model.decompress(d) for d in alone]

    def step(m, i):
        data = m.compress(sweeps[i])
        return data, m.decompress(data)
    with FramePipeline(model, depth=2) as pipe:
        assert pipe.models[1] is not model and pipe.models[1].block_dec_recurrent._described() is not model.block_dec_recurrent._described()
        out = pipe.map(step, [0, 1, 2, 3, 0, 1])
    for i, (data, rec) in zip([0, 1, 2, 3, 0, 1], out):
        assert data == alone[i] and torch.equal(rec, recs[i])
