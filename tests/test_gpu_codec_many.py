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
