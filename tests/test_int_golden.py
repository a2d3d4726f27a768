"""The integer LiDAR codec against tests/golden/codec_int.json: streams written by the REFERENCE's model code
(lib/int_sparse_conv/cuda_ops.py + models/convolutional/lossl_coord_int/model.py) and rANS coder, executed on the CPU by
tests/golden/make_golden.py over a stand-in for the CUDA extension (see its docstring for what the stand-in restates).
Everything is integer arithmetic: the oracle must reproduce the streams byte for byte on any machine."""
import hashlib
import json
import os

import numpy as np
import pytest

from fastpcc_amd.codecs.lossl_coord_int import Config, Model
from fastpcc_amd.codecs.lossl_coord_int.init_random import randomize_
from oracle.codec_int import OracleInt

with open(os.path.join(os.path.dirname(__file__), 'golden', 'codec_int.json')) as f:
    RUNS = json.load(f)['runs']


def model_of(run, device='cpu'):
    cfg = Config(**run['config'])
    model = Model(cfg, device)
    randomize_(model, run['seed'])
    return cfg, model


@pytest.mark.parametrize('run', RUNS, ids=[r['label'] for r in RUNS])
def test_oracle_reproduces_the_reference_stream(run):
    cfg, model = model_of(run)
    assert [[k, list(v.shape)] for k, v in model.state_dict().items()] == run['state_dict_keys']
    xyz = np.array(run['xyz'], dtype=np.int64)
    coords = np.concatenate((np.zeros((len(xyz), 1), np.int64), xyz), 1)
    ref_stream = bytes.fromhex(run['stream_hex'])
    oracle = OracleInt(model.state_dict(), cfg)
    assert oracle.compress(coords) == ref_stream
    rec = oracle.decompress(ref_stream)
    assert hashlib.sha256(np.ascontiguousarray(rec.astype(np.int32)).tobytes()).hexdigest() == run['recon_sha256']
    assert sorted(map(tuple, rec.tolist())) == sorted(map(tuple, xyz.tolist()))
