"""Several independent clouds through ONE traversal of the networks (`compress_many` / `decompress_many`, and the partition lists of
/root/reference/models/convolutional/lossy_coord_v2/model.py:247-256,277-288 built on them): every cloud's stream must be the BYTES
`compress` writes for that cloud alone, and decode to the same points -- including clouds on either side of PAD_MIN_ROWS (where the
summation order of the narrow layers depends on the cloud's own row count), and the clouds of the chain-order reference runs, whose
bytes come from executed reference model code (tests/golden/codec_v2_chain.json)."""
import hashlib

import numpy as np
import pytest
import torch

from util import batched, enliven, surface_cloud

pytestmark = pytest.mark.gpu


def _dev(xyz, shift=(0, 0, 0)):
    return torch.from_numpy(batched(xyz) + np.array([0, *shift])).to(torch.int32).cuda()


@pytest.fixture(scope='module')
def model():
    from fastpcc_amd.codecs.lossy_coord_v2 import Model
    from fastpcc_amd.codecs.lossy_coord_v2.model_config import baseline_r1
    torch.manual_seed(0)
    m = Model(baseline_r1())
    enliven(m, 0)
    return m.cuda().eval()


@pytest.fixture(scope='module')
def clouds():
    rng = np.random.default_rng(3)
    tiny = np.unique(rng.integers(0, 12, (60, 3)), axis=0)
    return [_dev(surface_cloud(1, 64, 8000)),                      # every level below PAD_MIN_ROWS, the decoder's 8 N candidates above
            _dev(surface_cloud(2, 256, 90000), (5, 0, 9)),         # levels on both sides of it
            _dev(tiny, (100, 200, 300)),                           # everything far below
            _dev(surface_cloud(4, 128, 30000))]


def _same_points(a: torch.Tensor, b: torch.Tensor) -> bool:
    return a.shape == b.shape and bool((a == b).all())


def test_every_stream_is_the_single_cloud_stream(model, clouds):
    from fastpcc_amd import engine as ME
    alone = [model.compress(c) for c in clouds]
    recs = [model.decompress(s) for s in alone]
    for c, r in zip(clouds, recs):
        assert r.shape[0] == c.shape[0]
    for pick in ([0, 1, 2, 3], [2, 1], [3, 0, 2], [1, 1]):
        many = model.compress_many([clouds[i] for i in pick])
        assert len(many) == len(pick)
        for i, s in zip(pick, many):
            assert s == alone[i], f'cloud {i} of batch {pick}: stream differs from the one coded alone'
        back = model.decompress_many(many)
        for i, r in zip(pick, back):
            assert _same_points(r, recs[i]), f'cloud {i} of batch {pick}: decoded points differ'
    ME.clear_global_coordinate_manager()


def test_mixed_pad_plans_were_exercised(model, clouds):
    """the batch above holds clouds on both sides of PAD_MIN_ROWS on the same layer: the engine must have taken its two-form path"""
    from fastpcc_amd import engine as ME
    calls = []
    orig = ME._ConvBase._forward_fused

    def spy(self, x, cm, src, coordinates, act, clip, plan_rows):
        calls.append((self.in_channels, self.out_channels, self.ks, plan_rows, src.n))
        return orig(self, x, cm, src, coordinates, act, clip, plan_rows)
    ME._ConvBase._forward_fused = spy
    try:
        model.decompress_many(model.compress_many([clouds[0], clouds[2]]))
    finally:
        ME._ConvBase._forward_fused = orig
    forced = {(c[0], c[1], c[2]) for c in calls if c[3] == ME.PAD_MIN_ROWS and c[4] != ME.PAD_MIN_ROWS}
    unforced = {(c[0], c[1], c[2]) for c in calls if c[3] == 0}
    assert forced & unforced, 'no layer ran in both forms'


def test_a_single_cloud_list_is_the_plain_call(model, clouds):
    assert model.compress_many([clouds[0]]) == [model.compress(clouds[0])]


def test_partition_lists_go_through_grouped_traversals(model, clouds, monkeypatch):
    want = b''.join(len(s).to_bytes(3, 'little') + s for s in (model.compress(c) for c in clouds))
    whole = torch.cat(clouds)
    sizes = [c.shape[0] for c in clouds]
    for cap, groups in ((10 ** 9, [[0, 1, 2, 3]]), (sum(sizes[:3]), [[0, 1, 2], [3]]), (1, [[0], [1], [2], [3]])):   # one traversal / two / four
        monkeypatch.setattr(type(model), 'MANY_MAX_VOXELS', cap)
        assert model._groups(sizes) == groups
        blob = model.compress_partitions([whole, *clouds])
        assert blob == want
        rec = model.decompress_partitions(blob)
        assert rec.shape[0] == sum(c.shape[0] for c in clouds)


def test_encoder_and_decoder_may_group_a_partition_list_differently(model, clouds, monkeypatch):
    """compress_partitions groups by input voxel counts, decompress_partitions by a proxy read from the headers: the two sides may
    batch DIFFERENT sets of clouds together.  That is sound only because a cloud's activations do not depend on what shares its batch
    (every output row is its own summation chain; the row-count rules -- PAD_MIN_ROWS, top-k, headers -- look at the cloud's own rows:
    include/fpcc_hip.h at fpcc_conv_f32_order_ex, engine._pad_plan).  Encode under every grouping, decode under every other one."""
    whole = torch.cat(clouds)
    sizes = [c.shape[0] for c in clouds]
    caps = (10 ** 9, sum(sizes[:3]), sizes[0] + sizes[1], 1)
    monkeypatch.setattr(type(model), 'MANY_MAX_VOXELS', 10 ** 9)
    want_blob = model.compress_partitions([whole, *clouds])
    want_rec = model.decompress_partitions(want_blob)
    for enc_cap in caps:
        monkeypatch.setattr(type(model), 'MANY_MAX_VOXELS', enc_cap)
        blob = model.compress_partitions([whole, *clouds])
        assert blob == want_blob
        for dec_cap in caps:
            if dec_cap == enc_cap:
                continue
            monkeypatch.setattr(type(model), 'MANY_MAX_VOXELS', dec_cap)
            assert _same_points(model.decompress_partitions(blob), want_rec), (enc_cap, dec_cap)


def _chain_runs():
    import json
    import os
    with open(os.path.join(os.path.dirname(__file__), 'golden', 'codec_v2_chain.json')) as f:
        g = json.load(f)
    from fastpcc_amd.codecs.lossy_coord_v2.model_config import ModelConfig
    keys = {f.name for f in __import__('dataclasses').fields(ModelConfig)}
    return g['numerics_version'], [dict(r, config={k: v for k, v in r['config'].items() if k in keys}) for r in g['runs']]


@pytest.mark.parametrize('run', _chain_runs()[1], ids=[r['label'] for r in _chain_runs()[1]])
def test_batched_streams_equal_the_reference_runs_in_chain_order(run):
    """the cloud of a chain-order reference run coded INSIDE a batch (first, last, beside a much larger cloud): its stream is the
    reference run's bytes, and decodes to the reference run's points"""
    from fastpcc_amd import hipops
    from fastpcc_amd.codecs.lossy_coord_v2 import Model
    from fastpcc_amd.codecs.lossy_coord_v2.model_config import ModelConfig
    assert _chain_runs()[0] == hipops.numerics_version(), 'numerics version bumped: regenerate codec_v2_chain.json'
    cfg = ModelConfig(**{k: tuple(v) if isinstance(v, list) else v for k, v in run['config'].items()})
    torch.manual_seed(0)
    model = Model(cfg)
    enliven(model, run['seed'])
    model = model.cuda().eval()
    mine = _dev(np.array(run['xyz'], dtype=np.int32))
    want = bytes.fromhex(run['stream_hex'])
    other, big = _dev(surface_cloud(9, 64, 6000), (1, 2, 3)), _dev(surface_cloud(10, 256, 80000))
    for batch, at in (([mine, other], 0), ([big, other, mine], 2)):
        streams = model.compress_many(batch)
        assert streams[at] == want
        rec = model.decompress_many(streams)[at].cpu().numpy().astype(np.int64)
        keys = np.sort((rec[:, 0] << 42) | (rec[:, 1] << 21) | rec[:, 2])
        assert len(rec) == run['recon_points'] and hashlib.sha256(keys.tobytes()).hexdigest() == run['recon_sha256']


# ---- the hand-over fallback of the encoder's coder pool ------------------------------------------------------------------------------
def test_a_failed_pool_job_is_coded_again_after_a_synchronise(model, clouds, monkeypatch):
    """a coder job that reports a refusal (what a job that read its inputs before they had landed would do) is repeated inline once
    every copy is complete: same bytes, a warning, and the process-wide counter bench.py reports"""
    from fastpcc_amd.coder_pool import CoderPool
    from fastpcc_amd.codecs.geo_lossl_em import GeoLosslessEntropyModel
    want = model.compress_many([clouds[0], clouds[3]])
    real_wait = CoderPool.wait
    fired = []

    def failing_wait(self):
        out = real_wait(self)
        if not fired and out:
            fired.append(1)
            raise RuntimeError('libfpcc_host: invalid argument')
        return out
    monkeypatch.setattr(CoderPool, 'wait', failing_wait)
    before = GeoLosslessEntropyModel.handover_retries
    with pytest.warns(UserWarning, match='succeeded when repeated'):
        got = model.compress_many([clouds[0], clouds[3]])
    assert fired and got == want and GeoLosslessEntropyModel.handover_retries == before + 1
    monkeypatch.setattr(CoderPool, 'wait', real_wait)
    assert model.compress(clouds[0]) == want[0]


# ---- the colour codec ------------------------------------------------------------------------------------------------------------------
def _colors(xyz, seed):
    rng = np.random.default_rng(seed)
    base = 127 + 90 * np.stack((np.sin(xyz[:, 0] / 9.0), np.cos(xyz[:, 1] / 7.0), np.sin((xyz[:, 2] + xyz[:, 0]) / 11.0)), 1)
    return torch.from_numpy(np.clip(base + rng.normal(0, 8, base.shape), 0, 255).astype(np.uint8)).cuda()


def test_colour_streams_are_the_single_cloud_streams():
    from fastpcc_amd.codecs.lossy_coord_lossy_color import Model
    from fastpcc_amd.codecs.lossy_coord_lossy_color.model_config import baseline_r1
    torch.manual_seed(0)
    m = Model(baseline_r1())
    enliven(m, 3, gain=2.3)
    m = m.cuda().eval()
    raw = [surface_cloud(7, 64, 16000), surface_cloud(8, 256, 70000), surface_cloud(9, 32, 1500)]
    xyz = [_dev(r, s) for r, s in zip(raw, ((2, 9, 0), (0, 0, 0), (40, 40, 40)))]
    rgb = [_colors(r, i) for i, r in enumerate(raw)]
    alone = [m.compress(c, f) for c, f in zip(xyz, rgb)]
    recs = [m.decompress(s) for s in alone]
    for pick in ([0, 1, 2], [2, 0], [1, 2]):
        many = m.compress_many([xyz[i] for i in pick], [rgb[i] for i in pick])
        for i, s in zip(pick, many):
            assert s == alone[i], f'coloured cloud {i} of batch {pick}: stream differs from the one coded alone'
        for i, (c, f) in zip(pick, m.decompress_many(many)):
            assert _same_points(c, recs[i][0]) and _same_points(f, recs[i][1]), f'coloured cloud {i} of batch {pick}: decode differs'
    blob = m.compress_partitions([torch.cat(xyz), *xyz], [torch.cat(rgb), *rgb])
    assert blob == b''.join(len(s).to_bytes(3, 'little') + s for s in alone)
    c, f = m.decompress_partitions(blob)
    assert c.shape[0] == sum(r[0].shape[0] for r in recs) == f.shape[0]


def _colour_chain_runs():
    import json
    import os
    with open(os.path.join(os.path.dirname(__file__), 'golden', 'codec_color_chain.json')) as f:
        g = json.load(f)
    return g['numerics_version'], g['runs']


@pytest.mark.parametrize('run', _colour_chain_runs()[1], ids=[r['label'] for r in _colour_chain_runs()[1]])
def test_batched_colour_streams_equal_the_reference_runs_in_chain_order(run):
    from dataclasses import fields
    from fastpcc_amd import hipops
    from fastpcc_amd.codecs.lossy_coord_lossy_color import Model
    from fastpcc_amd.codecs.lossy_coord_lossy_color.model_config import ModelConfig
    assert _colour_chain_runs()[0] == hipops.numerics_version(), 'numerics version bumped: regenerate codec_color_chain.json'
    known = {f.name for f in fields(ModelConfig)}
    cfg = ModelConfig(**{k: tuple(v) if isinstance(v, list) else v for k, v in run['config'].items() if k in known})
    torch.manual_seed(0)
    m = Model(cfg)
    enliven(m, run['seed'], **({} if run.get('gain') is None else {'gain': run['gain']}))
    m = m.cuda().eval()
    mine = _dev(np.array(run['xyz'], dtype=np.int32))
    mine_rgb = torch.from_numpy(np.array(run['color'], dtype=np.uint8)).cuda()
    other_raw = surface_cloud(11, 128, 40000)
    other, other_rgb = _dev(other_raw, (3, 1, 2)), _colors(other_raw, 5)
    want = bytes.fromhex(run['stream_hex'])
    for batch, colours, at in (([mine, other], [mine_rgb, other_rgb], 0), ([other, mine], [other_rgb, mine_rgb], 1)):
        streams = m.compress_many(batch, colours)
        assert streams[at] == want
        rec_xyz, rec_rgb = m.decompress_many(streams)[at]
        assert rec_xyz.cpu().numpy().tolist() == run['recon_xyz']
        assert rec_rgb.cpu().numpy().astype(int).tolist() == run['recon_rgb']


# ---- edges -----------------------------------------------------------------------------------------------------------------------------
def test_edges_of_the_batch_interface(model, clouds):
    """empty lists and host tensors are refused; a one-voxel cloud, a batch of many small clouds and repeated clouds code like any other"""
    with pytest.raises(RuntimeError):
        model.compress_many([])
    with pytest.raises(RuntimeError):
        model.compress_many([clouds[0], clouds[2].cpu()])
    one = _dev(np.array([[7, 7, 7]]), (5, 6, 7))
    rng = np.random.default_rng(12)
    small = [_dev(np.unique(rng.integers(0, 40, (300 + 50 * i, 3)), axis=0), (i, 2 * i, 3 * i)) for i in range(9)]
    batch = [one, *small, clouds[0], one]
    many = model.compress_many(batch)
    assert many[0] == many[-1] == model.compress(one)
    for c, s in zip(batch, many):
        assert s == model.compress(c)
    back = model.decompress_many(many)
    assert [b.shape[0] for b in back] == [model.decompress(s).shape[0] for s in many]
    assert back[0].cpu().tolist() == [[12, 13, 14]]                          # the one voxel, offset restored
    # the streams of a batch decode one at a time, and single-cloud streams decode as a batch
    assert _same_points(model.decompress(many[3]), back[3])
    assert _same_points(model.decompress_many([model.compress(c) for c in batch[:3]])[2], back[2])


def test_an_exception_inside_a_frame_leaves_the_coder_pool_idle(model, clouds, monkeypatch):
    """something raises in the middle of compress / decompress while jobs of the coder pool are queued or running (encoder jobs waiting
    for their flags, the background residual decode): the frame releases and waits for them before the exception leaves it -- the next
    frame finds an idle pool and its results are the usual ones"""
    from fastpcc_amd import hipops
    batch = [clouds[0], clouds[3]]
    want = model.compress_many(batch)
    want_points = [p.shape[0] for p in model.decompress_many(want)]
    real = hipops.logit_to_prob16
    for phase in ('compress', 'decompress'):
        calls = []

        def flaky(x):
            calls.append(1)
            if len(calls) == 3:
                raise RuntimeError('injected')
            return real(x)
        monkeypatch.setattr(hipops, 'logit_to_prob16', flaky)
        with pytest.raises(RuntimeError, match='injected'):
            model.compress_many(batch) if phase == 'compress' else model.decompress_many(want)
        monkeypatch.setattr(hipops, 'logit_to_prob16', real)
        pool = model.em_lossless_based._overlap[torch.device('cuda', torch.cuda.current_device())]['pool']
        assert pool.wait() == []                                              # nothing pending, no stale error
        assert model.compress_many(batch) == want
        assert [p.shape[0] for p in model.decompress_many(want)] == want_points


def _partition_runs():
    import json
    import os
    with open(os.path.join(os.path.dirname(__file__), 'golden', 'codec_v2_partitions_chain.json')) as f:
        g = json.load(f)
    from fastpcc_amd.codecs.lossy_coord_v2.model_config import ModelConfig
    keys = {f.name for f in __import__('dataclasses').fields(ModelConfig)}
    return g['numerics_version'], [dict(r, config={k: v for k, v in r['config'].items() if k in keys}) for r in g['runs']]


@pytest.mark.parametrize('run', _partition_runs()[1], ids=[r['label'] for r in _partition_runs()[1]])
def test_partition_lists_equal_the_reference_runs_in_chain_order(run):
    """STRICT, for the list path itself: the reference's compress_partitions / decompress_partitions (its own model code, clouds coded one
    after the other, baseline_r1.yaml at its real widths, documented summation orders) wrote `blob`; the product's compress_partitions
    -- ONE traversal over all clouds of the list -- must write the same bytes and decode them to the same points, cloud by cloud"""
    from fastpcc_amd import hipops
    from fastpcc_amd.codecs.lossy_coord_v2 import Model
    from fastpcc_amd.codecs.lossy_coord_v2.model_config import ModelConfig
    assert _partition_runs()[0] == hipops.numerics_version(), 'numerics version bumped: regenerate codec_v2_partitions_chain.json'
    cfg = ModelConfig(**{k: tuple(v) if isinstance(v, list) else v for k, v in run['config'].items()})
    torch.manual_seed(0)
    model = Model(cfg)
    enliven(model, run['seed'])
    model = model.cuda().eval()
    parts = [_dev(np.array(p, dtype=np.int32)) for p in run['parts']]
    want = bytes.fromhex(run['blob_hex'])
    calls = []
    real = model.compress_many
    model.compress_many = lambda clouds: (calls.append(len(clouds)), real(clouds))[1]
    blob = model.compress_partitions([torch.cat(parts), *parts])
    assert calls == [len(parts)], 'the list was not coded in one traversal'
    assert blob == want
    rec = model.decompress_partitions(want).cpu().numpy().astype(np.int64)
    at = 0
    for n_i, sha in zip(run['recon_points'], run['recon_sha256']):
        part = rec[at: at + n_i]
        keys = np.sort((part[:, 0] << 42) | (part[:, 1] << 21) | part[:, 2])
        assert hashlib.sha256(keys.tobytes()).hexdigest() == sha
        at += n_i
    assert at == len(rec)


def test_colour_partition_lists_equal_the_reference_run_in_chain_order():
    """the colour codec's list path: the reference's compress_partitions (clouds one after the other) against the product's (one
    traversal): same bytes, same points and colours cloud by cloud (codec_color_partitions_chain.json)"""
    import json
    import os
    from dataclasses import fields
    from fastpcc_amd import hipops
    from fastpcc_amd.codecs.lossy_coord_lossy_color import Model
    from fastpcc_amd.codecs.lossy_coord_lossy_color.model_config import ModelConfig
    with open(os.path.join(os.path.dirname(__file__), 'golden', 'codec_color_partitions_chain.json')) as f:
        g = json.load(f)
    assert g['numerics_version'] == hipops.numerics_version(), 'numerics version bumped: regenerate codec_color_partitions_chain.json'
    known = {f.name for f in fields(ModelConfig)}
    for run in g['runs']:
        cfg = ModelConfig(**{k: tuple(v) if isinstance(v, list) else v for k, v in run['config'].items() if k in known})
        torch.manual_seed(0)
        m = Model(cfg)
        enliven(m, run['seed'], gain=run['gain'])
        m = m.cuda().eval()
        xyz = [_dev(np.array(x, dtype=np.int32)) for x in run['xyz']]
        rgb = [torch.from_numpy(np.array(c, dtype=np.uint8)).cuda() for c in run['color']]
        want = bytes.fromhex(run['blob_hex'])
        assert m.compress_partitions([torch.cat(xyz), *xyz], [torch.cat(rgb), *rgb]) == want
        pos, streams = 0, []
        while pos != len(want):
            length = int.from_bytes(want[pos:pos + 3], 'little')
            streams.append(want[pos + 3: pos + 3 + length])
            pos += 3 + length
        for (c, f), part in zip(m.decompress_many(streams), run['parts']):
            assert c.cpu().numpy().tolist() == part['recon_xyz']
            assert f.cpu().numpy().astype(int).tolist() == part['recon_rgb']
