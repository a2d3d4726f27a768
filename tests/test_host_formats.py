"""Framing helpers and Morton keys against golden vectors from the reference's Python code."""
import hashlib
import io
import json
import os

import numpy as np
import pytest

from fastpcc_amd.bitstream import BytesListUtils
from oracle import codec_v2 as ocv2
from oracle import coords as oc
from oracle import lib as oracle_lib


def test_byteslist_golden(golden_dir):
    rng = np.random.default_rng(11)      # same generator state as tests/golden/make_golden.py:make_byteslist
    with open(os.path.join(golden_dir, 'byteslist.json')) as f:
        cases = json.load(f)
    for case in cases:
        strings = [bytes(rng.integers(0, 256, n, dtype=np.uint8)) for n in case['lengths']]
        assert hashlib.sha256(b''.join(strings)).hexdigest() == case['seed_strings_sha']
        for impl in ('product', 'oracle'):
            blob = BytesListUtils.concat_bytes_list(strings) if impl == 'product' else ocv2.concat_strings(strings)
            assert blob[: len(blob) - sum(case['lengths'])].hex() == case['head'], impl
            assert blob.endswith(b''.join(strings))
            back = BytesListUtils.split_bytes_list(blob, len(strings)) if impl == 'product' else \
                ocv2.split_strings(io.BytesIO(blob), len(strings))
            assert back == strings


def test_byteslist_rejects_bad_input():
    with pytest.raises(ValueError):
        BytesListUtils.concat_bytes_list([b'only one'])
    with pytest.raises(ValueError):
        BytesListUtils.split_bytes_list(None, 2, None)


def test_morton_golden(golden_dir):
    with open(os.path.join(golden_dir, 'morton.json')) as f:
        g = json.load(f)
    xyz = np.array(g['xyz'], dtype=np.int64)
    for tag, keys in g['keys'].items():
        order, inv = tag.split('|')
        assert oc.morton_encode(xyz, order, bool(int(inv))).tolist() == keys
    assert oc.morton_encode(np.array([[3, 5, 7]]))[0] == 431
    assert oc.morton_encode(np.array([[3, 5, 7]]), inverse=True)[0] == 239


def test_exp_lut_checksum(golden_dir):
    """the oracle regenerates the reference's hard-coded exponent table (softmax.cu:18-20) bit for bit"""
    import ctypes as C
    with open(os.path.join(golden_dir, 'explut.json')) as f:
        g = json.load(f)
    lut = np.zeros(g['n'], dtype='<i4')
    oracle_lib().orc_exp_lut(lut.ctypes.data_as(C.c_void_p))
    assert hashlib.sha256(lut.tobytes()).hexdigest() == g['sha256']
    assert lut[:8].tolist() == g['head'] and lut[-8:].tolist() == g['tail']
