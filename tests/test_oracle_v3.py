"""The v3 oracle on its own (CPU): round trips, the two evaluations of the convolution sum agree, side information."""
import numpy as np
import pytest
import torch

from fastpcc_amd.codecs.lossy_coord_v3 import Config, Model
from fastpcc_amd.codecs.lossy_coord_v3.init_random import randomize_
from fastpcc_amd.synthetic import batched, surface_cloud
from oracle.codec_v3 import OracleV3, histogram_cdf, quantize_pmf, top_children


def _weights(cfg, seed):
    model = Model(cfg)
    randomize_(model, seed)
    return model.state_dict()


def _rows(a):
    return sorted(map(tuple, np.asarray(a).tolist()))


@pytest.mark.parametrize('kw', [dict(num_latents=(0, 1, 2), lossl_geo_upsample=(1, 1, 1)),
                                dict(num_latents=(0, 0, 2, 2, 0), lossl_geo_upsample=(0, 1, 1, 1, 1)),
                                dict(num_latents=(0, 0, 2, 2, 0), lossl_geo_upsample=(0, 0, 1, 1, 1))])
def test_round_trip_and_modes(kw):
    cfg = Config(channels=16, max_stride=64, **kw)
    weights = _weights(cfg, 1)
    xyz = surface_cloud(3, 64, 3000) + np.array([2, 9, 0])
    coords = batched(xyz).astype(np.int64)
    n_lossy = next((i for i, v in enumerate(cfg.lossl_geo_upsample) if v), len(cfg.lossl_geo_upsample))
    lo = xyz.min(0)                                              # octree cells are aligned to the cloud's minimum corner
    out = {}
    for mode in ('mm', 'chain'):
        o = OracleV3(weights, cfg, conv=mode)
        data = o.compress(coords)
        rec = o.decompress(data)
        out[mode] = (data, dict(o.trace))
        assert [int.from_bytes(data[i:i + 2], 'little') for i in (0, 2, 4)] == xyz.min(0).tolist()
        assert _rows(np.unique((rec - lo) >> n_lossy, axis=0)) == _rows(np.unique((xyz - lo) >> n_lossy, axis=0))
        if n_lossy == 0:
            assert _rows(rec) == _rows(xyz)
        else:
            assert [int.from_bytes(data[8 + 3 * i: 11 + 3 * i], 'little') for i in range(n_lossy)] == \
                   [len(np.unique((xyz - lo) >> i, axis=0)) for i in range(n_lossy)]
            # top-k keeps at most the level's point count, every voxel keeps its best child on top of that (random weights:
            # the two sets overlap little)
            assert 0.5 * len(xyz) <= len(rec) <= len(xyz) + 2 * len(np.unique((xyz - lo) >> 1, axis=0))
    ta, tb = out['mm'][1], out['chain'][1]
    for k in ta:
        if k.startswith('latent'):
            assert np.mean(ta[k] != tb[k]) < 0.02
        if k.startswith('symbols'):
            assert (ta[k] == tb[k]).all()
    assert abs(len(out['mm'][0]) - len(out['chain'][0])) <= 0.02 * len(out['mm'][0])


def test_histogram_cdf_and_quantizer():
    v = np.array([0, 0, 0, 5, 5, 2, 7, 7, 7, 7])
    cdf = histogram_cdf(v)
    assert cdf.dtype == np.uint16 and len(cdf) == 8 and cdf[-1] == 65535
    f = np.diff(np.concatenate(([0], cdf.astype(np.int64))))
    assert (f >= 1).all() and f[7] > f[0] > f[5] > f[1]
    rows = quantize_pmf(np.zeros((2, 255), np.float32), True)
    assert rows.shape == (2, 255) and (np.diff(rows.astype(np.int64), axis=1) >= 1).all() and (rows[:, -1] == 65535).all()


def test_top_children_known_answers():
    x = np.array([[5, 1, 1, 1, 1, 1, 1, 1], [0, 0, 0, 0, 0, 0, 0, 0], [9, 8, 1, 1, 1, 1, 1, 1]], dtype=np.float32)
    m = top_children(x, 3)                                  # k = 21: the 21st smallest is 5 -> keep > 5, plus row maxima
    assert m[0].tolist() == [True] + [False] * 7
    assert m[1].all()                                       # a row of ties keeps all its maxima
    assert m[2].tolist() == [True, True] + [False] * 6
