"""lossy_coord_v3 against tests/golden/codec_v3.json: the REFERENCE's model code and rANS coder, executed on the CPU over a
functional torchsparse stand-in by tests/golden/make_golden.py (only the kernel-offset enumeration of the sparse
convolution is restated there; see its docstring).

Pinned exactly on any machine: module tree / state_dict layout, side-information tables, header bytes, the 255-ary symbols
and the order in which everything is coded, the histogram CDFs of the coarsest coordinates.  Latents, CDF rows, whole
streams and the decoder's reconstruction depend on fp32 GEMM and exp rounding: they are required to be IDENTICAL when this
machine's torch reproduces the generator's float probe (it does in the build container), and close otherwise."""
import hashlib
import json
import os

import numpy as np
import pytest
import torch

from fastpcc_amd.codecs.lossy_coord_v3 import Config, Model
from fastpcc_amd.codecs.lossy_coord_v3.init_random import randomize_
from oracle.codec_v3 import OracleV3, quantize_pmf

with open(os.path.join(os.path.dirname(__file__), 'golden', 'codec_v3.json')) as f:
    G = json.load(f)


def _same_float_behaviour() -> bool:
    g = torch.Generator().manual_seed(G['float_probe']['seed'])
    torch.tensor([5, 0, 0, 17, 1, 1, 250, 3])
    logits = torch.randn((3, 255), generator=g) * 3
    a, b = torch.randn((257, 40), generator=g), torch.randn((40, 24), generator=g)
    return hashlib.sha256(torch.mm(a, b).numpy().tobytes()).hexdigest() == G['float_probe']['mm_sha256'] and \
        hashlib.sha256(torch.softmax(logits, -1).numpy().tobytes()).hexdigest() == G['float_probe']['softmax_sha256']


@pytest.mark.parametrize('name', ['dense_r1', 'dense_r4', 'dense_r7'])
def test_state_dict_layout(name):
    ref = G['state_dict'][name]
    model = Model(Config(channels=ref['channels'], max_stride=ref['max_stride'], num_latents=tuple(ref['num_latents']),
                         lossl_geo_upsample=tuple(ref['lossl_geo_upsample'])))
    assert [[k, list(v.shape)] for k, v in model.state_dict().items()] == ref['keys']


def test_side_information_tables_and_quantizer():
    model = Model(Config())
    side = G['side_info']
    assert model.fea_side_info_cdf1[0, :4].tolist() == side['cdf1_head'] and model.fea_side_info_cdf1[0, -3:].tolist() == side['cdf1_tail']
    assert model.fea_side_info_cdf1.shape[1] == side['cdf1_len'] and model.fea_side_info_cdf2[0].tolist() == side['cdf2']
    assert model.bin2oct_kernel.tolist() == side['bin2oct_kernel'] and model.unfold_kernel[0].tolist() == side['unfold_kernel']
    o = OracleV3({}, Config())
    assert o.cdf1[0, :4].tolist() == side['cdf1_head'] and o.cdf2[0].tolist() == side['cdf2'] and o.unfold[0].tolist() == side['unfold_kernel']
    q = G['quantize_pmf']
    hist = np.array(q['hist'], dtype=np.float32)
    assert quantize_pmf((hist / hist.sum())[None], False)[0].tolist() == q['hist_cdf']
    assert Model.batch_quantize_pmf_torch((torch.tensor(q['hist']) / sum(q['hist']))[None], False)[0].tolist() == q['hist_cdf']
    rows = quantize_pmf(np.array(q['logits'], dtype=np.float32), True).astype(np.int64)
    assert np.abs(rows - np.array(q['logits_cdf'])).max() <= (0 if _same_float_behaviour() else 2)
    mine = Model.batch_quantize_pmf_torch(torch.tensor(q['logits'])).numpy()
    assert np.abs(mine - np.array(q['logits_cdf'])).max() <= (0 if _same_float_behaviour() else 2)


def test_bound_function():
    from fastpcc_amd.codecs.lossy_coord_v3.model import _Bound
    b = G['bound']
    x = torch.tensor(b['x'], requires_grad=True)
    y = _Bound.apply(x, torch.tensor(20.0))
    y.backward(torch.full_like(y, 0.5))
    assert y.tolist() == b['y'] and x.grad.tolist() == b['grad']


@pytest.mark.parametrize('run', G['runs'], ids=[r['label'] for r in G['runs']])
def test_reference_run(run):
    cfg = Config(**{k: tuple(v) if isinstance(v, list) else v for k, v in run['config'].items()})
    model = Model(cfg)
    randomize_(model, run['seed'])
    assert float(sum(p.detach().double().abs().sum() for n, p in model.named_parameters() if '.prior_' not in n)) == \
        pytest.approx(run['param_abs_sum'], rel=1e-12)     # same seeded weights on both sides (the priors are not read at test time)
    xyz = np.array(run['xyz'], dtype=np.int64)
    coords = np.concatenate((np.zeros((len(xyz), 1), np.int64), xyz), 1)
    ref_stream = bytes.fromhex(run['stream_hex'])
    exact = _same_float_behaviour()

    oracle = OracleV3(model.state_dict(), cfg, conv='mm')
    data = oracle.compress(coords)
    n_lossy = next((i for i, v in enumerate(cfg.lossl_geo_upsample) if v), len(cfg.lossl_geo_upsample))
    head = 8 + 3 * n_lossy
    assert data[:head] == ref_stream[:head]
    # what was coded, in coding order
    assert [s.astype(int).tolist() for s in oracle.coded['symbols']] == run['oct_symbols_in_coding_order']
    assert len(oracle.coded['fea']) == len(run['fea_in_coding_order'])
    for (cdf, values, lo), ref in zip(oracle.coded['fea'], run['fea_in_coding_order']):
        assert len(values) == len(ref['values'])
        same = np.mean((np.asarray(values) - (lo or 0)) == (np.array(ref['values']) - (ref['lo'] or 0)))
        assert same == 1.0 if exact or ref['lo'] is None else same > 0.97
        if exact or ref['lo'] is None:                       # the coarsest coordinates never depend on float arithmetic
            assert cdf.astype(int).tolist() == ref['cdf'] and lo == ref['lo']
    if exact:
        assert [hashlib.sha256(np.ascontiguousarray(r).tobytes()).hexdigest() for r in oracle.coded['rows']] == run['oct_cdf_sha256_in_coding_order']
        assert data == ref_stream
        assert oracle.decompress(ref_stream).tolist() == run['recon']
    else:
        assert abs(len(data) - len(ref_stream)) <= 0.02 * len(ref_stream) + 4
        assert abs(len(oracle.decompress(data)) - len(run['recon'])) <= 0.05 * len(run['recon'])
    if n_lossy == 0:
        assert sorted(map(tuple, run['recon'])) == sorted(map(tuple, xyz.tolist()))
