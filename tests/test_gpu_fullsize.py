"""The codecs at the sizes BASELINE.json names (cfg#2: 1 M voxels at 1024^3, cfg#3: a 64 x 2048 LiDAR sweep with 256
channels, cfg#4: 2 M coloured voxels at 2048^3).  The oracle cannot follow to these sizes in seconds everywhere, so what is
asserted are size-independent properties:

  * encode -> decode returns the coded point count (cfg#2 / #4; ties at the pruning threshold excepted, as in the reference)
    or the input set itself (cfg#3, lossless), the result has no duplicate voxels, and coding is deterministic;
  * the BYTES do not depend on any tuning: neighbour-pattern row order on / off, workgroup-tiled kernel with 128 / 64 / 32-row
    tiles, wave kernel with 1 / 2 / 4 column blocks per wave and with / without interleaved address arithmetic.  Every size-
    dependent dispatch (tile heights, LPT tile order and its 16 Ki-group cut-off, row-order threshold) is crossed here;
  * one comparison against the oracle at > 300 K voxels: identical symbols, probabilities within one LSB, identical
    reconstruction.
"""
import numpy as np
import pytest
import torch

from fastpcc_amd.synthetic import SCALE, batched, body_cloud, enliven, lidar_cloud

pytestmark = pytest.mark.gpu


def _key(xyz: np.ndarray) -> np.ndarray:
    xyz = xyz.astype(np.int64)
    return np.sort((xyz[:, 0] << 42) | (xyz[:, 1] << 21) | xyz[:, 2])


class _Tuning:
    """context: one setting of every result-neutral knob; restores the defaults"""

    def __init__(self, ops, ME, int_conv=None, *, row_order=True, wave=1, nbw=0, sb=1, tile=0, pointwise=32768, gnbw=0, w22=0, fold=102400):
        self.ops, self.ME, self.int_conv = ops, ME, int_conv
        self.want = dict(row_order=row_order, wave=wave, nbw=nbw, sb=sb, tile=tile, pointwise=pointwise, gnbw=gnbw, w22=w22, fold=fold)

    def __enter__(self):
        o, w = self.ops, self.want
        self.saved = [o.conv_set_tuning(k, v) for k, v in ((o.KNOB_WAVE_ON, w['wave']), (o.KNOB_WAVE_NBW, w['nbw']),
                                                           (o.KNOB_WAVE_SB, w['sb']), (o.KNOB_MFMA_TILE, w['tile']),
                                                           (o.KNOB_POINTWISE_ROWS, w['pointwise']), (o.KNOB_GROUPED_NBW, w['gnbw']),
                                                           (o.KNOB_WAVE22_ROWS, w['w22']), (o.KNOB_GROUPED_FOLD_ROWS, w['fold']))]
        self.saved_rows = self.ME.CoordinateManager.ROW_ORDER_MIN_ROWS
        if not w['row_order']:
            self.ME.CoordinateManager.ROW_ORDER_MIN_ROWS = 1 << 40
        if self.int_conv is not None:
            self.saved_int = self.int_conv.ROW_ORDER_MIN_ROWS
            if not w['row_order']:
                self.int_conv.ROW_ORDER_MIN_ROWS = 1 << 40
        return self

    def __exit__(self, *exc):
        o = self.ops
        for k, v in zip((o.KNOB_WAVE_ON, o.KNOB_WAVE_NBW, o.KNOB_WAVE_SB, o.KNOB_MFMA_TILE, o.KNOB_POINTWISE_ROWS, o.KNOB_GROUPED_NBW,
                         o.KNOB_WAVE22_ROWS, o.KNOB_GROUPED_FOLD_ROWS), self.saved):
            o.conv_set_tuning(k, v)
        self.ME.CoordinateManager.ROW_ORDER_MIN_ROWS = self.saved_rows
        if self.int_conv is not None:
            self.int_conv.ROW_ORDER_MIN_ROWS = self.saved_int


TUNINGS = [dict(row_order=False), dict(wave=0), dict(wave=0, tile=1), dict(wave=0, tile=2), dict(wave=0, tile=3),
           dict(wave=0, row_order=False, tile=2), dict(nbw=1), dict(nbw=2, sb=0), dict(nbw=4), dict(nbw=4, row_order=False),
           dict(pointwise=0), dict(pointwise=1), dict(pointwise=1, nbw=1, row_order=False),
           dict(gnbw=1), dict(gnbw=2), dict(gnbw=2, row_order=False), dict(w22=1),
           dict(fold=0), dict(fold=1), dict(fold=1, row_order=False)]     # persistent per-point kernel: never / always


@pytest.fixture(scope='module')
def v2():
    from fastpcc_amd import engine as ME, hipops
    from fastpcc_amd.codecs.lossy_coord_v2 import Model
    from fastpcc_amd.codecs.lossy_coord_v2.model_config import baseline_r1
    cfg = baseline_r1()
    torch.manual_seed(0)
    model = Model(cfg)
    enliven(model, 0)
    weights = {k: v.clone() for k, v in model.state_dict().items() if isinstance(v, torch.Tensor)}
    return cfg, model.cuda().eval(), weights, hipops, ME


def test_cfg2_one_million_voxels_round_trip_and_tuning_invariance(v2):
    cfg, model, _, ops, ME = v2
    xyz = body_cloud(1024, SCALE[1024], seed=2)
    assert abs(len(xyz) - 1_000_000) <= 10_000                       # cfg#2: 1 M voxels +- 1 %
    frame = torch.from_numpy(batched(xyz)).cuda()
    data = model.compress(frame)
    ME.clear_global_coordinate_manager()
    rec = model.decompress(data).cpu().numpy()
    ME.clear_global_coordinate_manager()
    n = len(xyz)
    assert n - max(16, n // 1000) <= len(rec) <= n                    # ties with the k-th value are dropped (kthvalue rule)
    keys = _key(rec)
    assert (np.diff(keys) > 0).all()                                  # no voxel twice
    assert rec.min() >= 0 and rec.max() < 1024
    assert int.from_bytes(data[6:9], 'little') == n                   # header carries the point count
    assert model.compress(frame) == data                              # deterministic
    ME.clear_global_coordinate_manager()
    for t in TUNINGS:
        with _Tuning(ops, ME, **t):
            again = model.compress(frame)
            ME.clear_global_coordinate_manager()
            assert again == data, f'bytes depend on the tuning {t}'
            rec2 = model.decompress(data).cpu().numpy()
            ME.clear_global_coordinate_manager()
            assert (_key(rec2) == keys).all(), f'reconstruction depends on the tuning {t}'


def test_v2_against_the_oracle_at_300k_voxels(v2):
    from fastpcc_amd.engine import summation_order as ME_order
    from oracle.codec_v2 import OracleV2
    cfg, model, weights, ops, ME = v2
    xyz = body_cloud(576, SCALE[1024], seed=5)
    assert len(xyz) > 300_000
    coords = batched(xyz)
    model.em_lossless_based.keep_symbols = True
    try:
        data = model.compress(torch.from_numpy(coords).cuda())
        sym = model.em_lossless_based.last_symbols
    finally:
        model.em_lossless_based.keep_symbols = False
    ME.clear_global_coordinate_manager()
    rec = model.decompress(data).cpu().numpy()
    ME.clear_global_coordinate_manager()
    o = OracleV2(weights, cfg, conv='chain', order_fn=ME_order)
    o.skip_unused_tail = True
    want = o.compress(coords.astype(np.int64))
    assert (sym['residual'].reshape(-1) == o.symbols['residual'].reshape(-1)).all()
    assert (sym['occupancy'].astype(bool) == np.concatenate(o.symbols['occupancy'])).all()
    p_gpu, p_cpu = sym['prob'].astype(np.int64), np.concatenate(o.symbols['prob']).astype(np.int64)
    assert (p_gpu == p_cpu).all()                                   # numerics version 3: specified logistic function, no tolerance
    assert data == want
    rec_o = o.decompress(want)                                       # same symbols -> the same reconstruction, voxel for voxel
    assert (_key(np.asarray(rec_o)[:, -3:]) == _key(rec)).all()


def test_cfg3_lidar_sweep_256_channels_lossless_and_tuning_invariance():
    from fastpcc_amd import engine as ME, hipops as ops, int_sparse_conv
    from fastpcc_amd.codecs.lossl_coord_int import Config, Model
    from fastpcc_amd.codecs.lossl_coord_int.init_random import randomize_
    cfg = Config()                                                    # channels 256, the reference's defaults
    assert cfg.channels == 256
    model = Model(cfg, 'cuda')
    randomize_(model, 5)
    model = model.cuda().eval()
    xyz = lidar_cloud()                                               # 64 beams x 2048 azimuths, 16-bit voxels (cfg#3)
    assert 100_000 < len(xyz) < 130_000
    frame = torch.from_numpy(batched(xyz)).cuda()
    data = model.compress(frame)
    rec = model.decompress(data).cpu().numpy()
    assert (_key(rec) == _key(xyz)).all()                             # lossless
    assert model.compress(frame) == data
    perm = torch.randperm(len(xyz), generator=torch.Generator().manual_seed(0)).cuda()
    assert model.compress(frame[perm]) == data                        # input order is irrelevant
    with _Tuning(ops, ME, int_sparse_conv, row_order=False):
        assert model.compress(frame) == data, 'bytes depend on the row order of the int8 convolution'
        assert (_key(model.decompress(data).cpu().numpy()) == _key(xyz)).all()
    # the level-per-call traversal (fpcc_int_level_*) against the module-by-module one, and a batch of sweeps against single sweeps
    from fastpcc_amd.codecs.lossl_coord_int import model as int_model
    assert int_model.FAST_LEVELS
    int_model.FAST_LEVELS = False
    try:
        assert model.compress(frame) == data, 'bytes depend on the traversal path'
        assert (_key(model.decompress(data).cpu().numpy()) == _key(xyz)).all()
    finally:
        int_model.FAST_LEVELS = True
    other = torch.from_numpy(batched(lidar_cloud(7))).cuda()
    many = model.compress_many([other, frame])
    assert many[1] == data and many[0] == model.compress(other)
    back = model.decompress_many(many)
    assert (_key(back[1].cpu().numpy()) == _key(xyz)).all() and back[0].shape[0] == other.shape[0]


def test_cfg4_two_million_coloured_voxels_round_trip_and_tuning_invariance():
    from fastpcc_amd import engine as ME, hipops as ops
    from fastpcc_amd.codecs.lossy_coord_lossy_color import Model
    from fastpcc_amd.codecs.lossy_coord_lossy_color.model_config import baseline_r1
    cfg = baseline_r1()
    torch.manual_seed(0)
    model = Model(cfg)
    enliven(model, 3, gain=2.3)
    model = model.cuda().eval()
    xyz = body_cloud(2048, SCALE[2048], seed=4)
    assert abs(len(xyz) - 2_000_000) <= 20_000
    rng = np.random.default_rng(1)
    base = 127 + 90 * np.stack((np.sin(xyz[:, 0] / 90.0), np.cos(xyz[:, 1] / 70.0), np.sin((xyz[:, 2] + xyz[:, 0]) / 110.0)), 1)
    rgb = np.clip(base + rng.normal(0, 8, base.shape), 0, 255).astype(np.uint8)
    frame, color = torch.from_numpy(batched(xyz)).cuda(), torch.from_numpy(rgb).cuda()
    data = model.compress(frame, color)
    ME.clear_global_coordinate_manager()
    rec_xyz, rec_rgb = model.decompress(data)
    ME.clear_global_coordinate_manager()
    rec_xyz, rec_rgb = rec_xyz.cpu().numpy(), rec_rgb.cpu().numpy()
    n = len(xyz)
    assert n - max(16, n // 1000) <= len(rec_xyz) <= n and rec_rgb.shape == (len(rec_xyz), 3)
    keys = _key(rec_xyz)
    assert (np.diff(keys) > 0).all()
    assert rec_rgb.min() >= 0 and rec_rgb.max() <= 255 and (rec_rgb == np.round(rec_rgb)).all()
    order = np.argsort((rec_xyz[:, 0].astype(np.int64) << 42) | (rec_xyz[:, 1].astype(np.int64) << 21) | rec_xyz[:, 2])
    assert model.compress(frame, color) == data
    ME.clear_global_coordinate_manager()
    for t in (dict(row_order=False), dict(wave=0), dict(wave=0, tile=1), dict(nbw=1), dict(nbw=4, sb=0)):
        with _Tuning(ops, ME, **t):
            assert model.compress(frame, color) == data, f'bytes depend on the tuning {t}'
            ME.clear_global_coordinate_manager()
            x2, c2 = model.decompress(data)
            ME.clear_global_coordinate_manager()
            x2, c2 = x2.cpu().numpy(), c2.cpu().numpy()
            o2 = np.argsort((x2[:, 0].astype(np.int64) << 42) | (x2[:, 1].astype(np.int64) << 21) | x2[:, 2])
            assert (x2[o2] == rec_xyz[order]).all() and (c2[o2] == rec_rgb[order]).all(), f'reconstruction depends on {t}'


def test_committed_stream_of_this_numerics_version_still_decodes(v2):
    """tests/golden/v2_stream.json was written by tools/make_gpu_golden.py on an MI355X: a later build that changes an
    order-selecting constant (include/fpcc_hip.h, 'Numerics version') decodes it to a different cloud -- and fails here
    instead of silently orphaning streams."""
    import hashlib, json, os
    from fastpcc_amd.synthetic import surface_cloud
    path = os.path.join(os.path.dirname(__file__), 'golden', 'v2_stream.json')
    if not os.path.exists(path):
        pytest.skip('no committed GPU stream yet (tools/make_gpu_golden.py)')
    cfg, model, _, ops, ME = v2
    with open(path) as f:
        g = json.load(f)
    assert g['numerics_version'] == ops.numerics_version(), 'numerics version bumped: regenerate the golden stream deliberately'
    data = bytes.fromhex(g['stream_hex'])
    rec = model.decompress(data).cpu().numpy()
    ME.clear_global_coordinate_manager()
    keys = _key(rec)
    assert len(rec) == g['decoded_voxels'] and hashlib.sha256(keys.tobytes()).hexdigest() == g['decoded_sha256']
    xyz = surface_cloud(11, 128, 90000)
    again = model.compress(torch.from_numpy(batched(xyz)).to(torch.int32).cuda())
    ME.clear_global_coordinate_manager()
    assert again == data, 'the encoder no longer writes the committed stream for the committed input'


@pytest.mark.parametrize('version', [1, 2])
def test_streams_of_earlier_numerics_versions_are_refused(v2, version):
    """tests/golden/v2_stream_numerics{1,2}.json are the streams the round-2 build (version 1: other summation orders for the
    multi-offset layers) and the round-3 build (version 2: probabilities from the device's expf) wrote for the same cloud and
    weights.  With the version byte in use a later build must refuse them; the version-1 bytes also differ from what this build writes
    (the activations differ, so do the coded symbols)."""
    import copy, json, os
    from fastpcc_amd.codecs.lossy_coord_v2 import Model
    cfg, model, _, ops, ME = v2
    with open(os.path.join(os.path.dirname(__file__), 'golden', f'v2_stream_numerics{version}.json')) as f:
        old = json.load(f)
    with open(os.path.join(os.path.dirname(__file__), 'golden', 'v2_stream.json')) as f:
        cur = json.load(f)
    assert old['numerics_version'] == version and ops.numerics_version() > version
    assert old['cloud'] == cur['cloud'] and old['weights'] == cur['weights']
    if version == 1:
        assert old['stream_hex'] != cur['stream_hex']
    cfg2 = copy.deepcopy(cfg)
    cfg2.numerics_version_in_header = True
    versioned = Model(cfg2)
    versioned.load_state_dict(model.state_dict())
    versioned = versioned.cuda().eval()
    with pytest.raises(ValueError, match=f'numerics version {version}'):
        versioned.decompress(bytes([version]) + bytes.fromhex(old['stream_hex']))
    ME.clear_global_coordinate_manager()
    # and the current stream, carried with its version byte, decodes
    rec = versioned.decompress(bytes([ops.numerics_version()]) + bytes.fromhex(cur['stream_hex']))
    ME.clear_global_coordinate_manager()
    assert len(rec) == cur['decoded_voxels']


def test_cfg3_sweeps_of_three_sizes_lossless_on_both_traversal_paths():
    """six LiDAR-like sweeps (28 K, 64 K, 113 K voxels; different seeds put their octree levels on both sides of the 8192-row switch
    between the offset-split and the tiled int8 convolution): lossless, the level-per-call path writes the module-by-module path's bytes,
    a batch writes the single sweeps' streams"""
    from fastpcc_amd.codecs.lossl_coord_int import Config, Model, model as int_model
    from fastpcc_amd.codecs.lossl_coord_int.init_random import randomize_
    model = Model(Config(), 'cuda')
    randomize_(model, 1)
    model = model.cuda().eval()
    frames, streams = [], []
    for s in range(6):
        beams, az = (64, 2048) if s % 3 == 0 else ((32, 1024) if s % 3 == 1 else (48, 1536))
        xyz = lidar_cloud(100 + s, beams=beams, azimuths=az)
        frame = torch.from_numpy(batched(xyz)).cuda()
        data = model.compress(frame)
        assert (_key(model.decompress(data).cpu().numpy()) == _key(xyz)).all(), f'sweep {s}: not lossless'
        int_model.FAST_LEVELS = False
        try:
            assert model.compress(frame) == data, f'sweep {s}: the traversal paths write different bytes'
        finally:
            int_model.FAST_LEVELS = True
        frames.append(frame)
        streams.append(data)
    assert model.compress_many(frames[:3]) == streams[:3] and model.compress_many(frames[3:]) == streams[3:]
