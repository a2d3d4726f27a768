"""Entropy coders against the golden vectors captured from the reference's C++ coders (tests/golden/rans.json, made by
tests/golden/make_golden.py), for BOTH the oracle restatement (oracle/rans.c) and the product's host library
(fastpcc_amd/csrc/host, through its C ABI).  Where oracle/_ref is present the reference itself is run as well.
Mirrors the reference's own self-tests (/root/reference/lib/entropy_models/rans_coder/__init__.py:9-96,
/root/reference/models/convolutional/lossy_coord_v3/rans_coder/__init__.py:28-63)."""
import json
import os
import sys

import numpy as np
import pytest

import oracle
from oracle import rans as orc
from fastpcc_amd import rans_coder as prod

IMPLS = {'oracle': orc, 'product': prod}


def _ref_modules():
    d = oracle.ref_dir()
    if not os.path.exists(os.path.join(d, 'rans_ext_cpp.so')):
        return None
    sys.path.insert(0, d)
    try:
        import rans_ext_cpp
        import simple_rans_ext_cpp
    except Exception:
        return None
    import types
    return types.SimpleNamespace(IndexedRansCoder=rans_ext_cpp.IndexedRansCoder,
                                 BinaryRansCoder=rans_ext_cpp.BinaryRansCoder,
                                 batched_pmf_to_quantized_cdf=rans_ext_cpp.batched_pmf_to_quantized_cdf,
                                 RansEncoder=simple_rans_ext_cpp.RansEncoder, RansDecoder=simple_rans_ext_cpp.RansDecoder)


@pytest.fixture(scope='module')
def G(golden_dir):
    with open(os.path.join(golden_dir, 'rans.json')) as f:
        return json.load(f)


@pytest.mark.parametrize('impl', list(IMPLS))
def test_pmf_to_cdf_golden(G, impl):
    M = IMPLS[impl]
    for case in G['cdf']:
        pmf = np.array(case['pmf'], dtype=np.float64)
        off = np.full(pmf.shape[0], case['offset_in'], dtype=np.int32)
        cdfs = M.batched_pmf_to_quantized_cdf(pmf, off, case['overflow'])
        assert [list(map(int, c)) for c in cdfs] == case['cdf']
        assert off.tolist() == case['offset_out']


@pytest.mark.parametrize('impl', list(IMPLS))
def test_reference_known_answer(impl):
    """the one known-answer assertion of the reference (rans_coder/__init__.py:72-77)"""
    M = IMPLS[impl]
    pmf = np.array([[0, 0, 0, 0], [1, 0, 0, 0], [0, 0, 0, 1], [2 ** -17, 1, 0, 0]], dtype=np.float64)
    off = np.zeros(4, dtype=np.int32)
    coder = M.IndexedRansCoder(True, 2, 100)
    coder.init_with_pmfs(pmf, off)
    assert coder.get_cdfs() == [[0, 1, 65536], [0, 65535, 65536], [0, 65535, 65536], [0, 65535, 65536]]
    assert off.tolist() == [4, 0, 3, 1]
    sym = np.array([[-2, -1], [0, 10]], dtype=np.int32)
    idx = np.array([[0, 1], [2, 2]], dtype=np.int32)
    dec = np.empty_like(sym)
    coder.decode_with_indexes(coder.encode_with_indexes(sym, idx), idx, dec)
    assert (dec == sym).all()


@pytest.mark.parametrize('impl', list(IMPLS))
def test_indexed_golden(G, impl):
    M = IMPLS[impl]
    for case in G['indexed']:
        coder = M.IndexedRansCoder(case['overflow'], 1)
        coder.init_with_quantized_cdfs(case['cdfs'], np.array(case['offsets'], dtype=np.int32))
        sym = np.array([case['symbols']], dtype=np.int32)
        idx = np.array([case['indexes']], dtype=np.int32)
        enc = coder.encode_with_indexes(sym, idx)[0] if case['with_indexes'] else coder.encode(sym)[0]
        assert enc.hex() == case['stream']
        dec = np.empty_like(sym)
        if case['with_indexes']:
            coder.decode_with_indexes([bytes.fromhex(case['stream'])], idx, dec)
        else:
            coder.decode([bytes.fromhex(case['stream'])], dec)
        assert (dec == sym).all()


@pytest.mark.parametrize('impl', list(IMPLS))
def test_binary_golden(G, impl):
    M = IMPLS[impl]
    for case in G['binary']:
        prob = np.array([case['prob']], dtype=np.uint32)
        bits = np.array([case['bits']], dtype=np.bool_)
        coder = M.BinaryRansCoder(1)
        assert coder.encode(bits, prob)[0].hex() == case['stream']
        dec = np.empty_like(bits)
        coder.decode([bytes.fromhex(case['stream'])], prob, dec)
        assert (dec == bits).all()


@pytest.mark.parametrize('impl', list(IMPLS))
def test_simple_golden(G, impl):
    M = IMPLS[impl]
    for case in G['simple']:
        enc = M.RansEncoder(1 << 20)
        if 'bin' in case:
            edge = np.array(case['bin']['edge'], dtype=np.uint16)
            bits = np.array(case['bin']['bits'], dtype=np.bool_)
            enc.encode_bin(edge, bits)
            assert enc.flush().hex() == case['stream']
            dec = M.RansDecoder()
            dec.flush(bytes.fromhex(case['stream']))
            got = np.zeros_like(bits)
            dec.decode_bin(edge, got)
            assert (got == bits).all()
            continue
        for blk in case['blocks']:
            enc.encode(np.array(blk['rows'], dtype=np.uint16), np.array(blk['symbols'], dtype=np.uint16))
        stream = enc.flush()
        assert stream.hex() == case['stream']
        dec = M.RansDecoder()
        dec.flush(stream)
        for blk in reversed(case['blocks']):          # LIFO: last pushed block is decoded first
            got = np.zeros(len(blk['symbols']), dtype=np.uint16)
            dec.decode(np.array(blk['rows'], dtype=np.uint16), got)
            assert got.tolist() == blk['symbols']


@pytest.mark.parametrize('impl', list(IMPLS))
def test_edge_cases(impl):
    M = IMPLS[impl]
    # empty symbol arrays still produce the 4-byte state
    coder = M.IndexedRansCoder(False, 1)
    coder.init_with_quantized_cdfs([[0, 1, 65536]], np.zeros(1, np.int32))
    assert coder.encode(np.zeros((1, 0), np.int32))[0].hex() == '00008000'
    assert M.BinaryRansCoder(1).encode(np.zeros((1, 0), np.bool_), np.zeros((1, 0), np.uint32))[0].hex() == '00008000'
    # extreme probabilities and large escape values round-trip
    prob = np.array([[1, 65535, 1, 65535, 32768]], dtype=np.uint32)
    bits = np.array([[1, 0, 0, 1, 1]], dtype=np.bool_)
    b = M.BinaryRansCoder(1)
    dec = np.empty_like(bits)
    b.decode(b.encode(bits, prob), prob, dec)
    assert (dec == bits).all()
    c = M.IndexedRansCoder(True, 1)
    c.init_with_pmfs(np.array([[.1, .2, .4, .2, .1]]), np.array([-2], np.int32))
    sym = np.array([[-2, 0, 2, 50, -50, 1, 2049, -2049, 1 << 20, -(1 << 20)]], dtype=np.int32)
    dec = np.empty_like(sym)
    c.decode(c.encode(sym), dec)
    assert (dec == sym).all()


def test_batch_units_are_independent():
    """batch_size > 1: one stream per unit (the reference parallelises them with OpenMP, rans_wrapper.cpp:110-111)"""
    rng = np.random.default_rng(5)
    for M in IMPLS.values():
        coder = M.IndexedRansCoder(False, 4, 100)
        coder.init_with_pmfs(np.array([[0, 0, 1], [1, 1, 2]], dtype=np.float64), np.zeros(2, np.int32))
        sym = np.array([[2, 1, 2, 0]] * 4, dtype=np.int32)
        dec = np.empty_like(sym)
        enc = coder.encode(sym)
        assert len(enc) == 4 and len(set(enc)) == 1
        coder.decode(enc, dec)
        assert (dec == sym).all()
        b = M.BinaryRansCoder(2, 100)
        bits = rng.integers(0, 2, (2, 100)).astype(np.bool_)
        prob = np.clip(np.round(rng.random((2, 100)) * 65536), 1, 65535).astype(np.uint32)
        got = np.empty_like(bits)
        b.decode(b.encode(bits, prob), prob, got)
        assert (got == bits).all()


def test_against_live_reference():
    """When the reference's own coders were built here (oracle/_ref), compare random streams three ways."""
    R = _ref_modules()
    if R is None:
        pytest.skip('oracle/_ref not built (needs /root/reference)')
    rng = np.random.default_rng(99)
    for _ in range(10):
        n = int(rng.integers(1, 3000))
        prob = np.clip(np.round(rng.beta(.3, .3, (1, n)) * 65536), 1, 65535).astype(np.uint32)
        bits = rng.random((1, n)) < prob / 65536
        want = R.BinaryRansCoder(1).encode(bits, prob)[0]
        assert orc.BinaryRansCoder(1).encode(bits, prob)[0] == want
        assert prod.BinaryRansCoder(1).encode(bits, prob)[0] == want
        hist = rng.integers(0, 50, int(rng.integers(2, 41))).astype(np.float64)
        hist[0] += 1
        sym = rng.choice(len(hist), size=(1, n), p=hist / hist.sum()).astype(np.int32) - 7
        streams = []
        for M in (R, orc, prod):
            c = M.IndexedRansCoder(False, 1)
            c.init_with_pmfs(hist[None].copy(), np.array([-7], np.int32))
            streams.append((c.get_cdfs(), c.encode(sym)[0]))
        assert streams[0] == streams[1] == streams[2]


def test_multi_stream_encoder_matches_single():
    from fastpcc_amd._native import host, host_check
    rng = np.random.default_rng(3)
    sizes = [0, 5, 1000, 64, 20000]
    n = sum(sizes)
    prob = np.clip(np.round(rng.random(n) * 65536), 1, 65535).astype(np.uint16)
    bits = (rng.random(n) < prob / 65536).astype(np.uint8)
    start = np.concatenate(([0], np.cumsum(sizes))).astype(np.int64)
    cap = 4 * max(sizes) + 64
    out = np.empty((len(sizes), cap), np.uint8)
    lens = np.zeros(len(sizes), np.int64)
    host_check(host().fpcc_rans_binary_encode_multi(bits.ctypes.data, prob.ctypes.data, start.ctypes.data, len(sizes),
                                                    out.ctypes.data, cap, lens.ctypes.data, 4))
    for s, size in enumerate(sizes):
        want = orc.BinaryRansCoder(1).encode(bits[start[s]:start[s + 1]].astype(bool)[None],
                                             prob[start[s]:start[s + 1]].astype(np.uint32)[None])[0]
        assert out[s, cap - lens[s]:].tobytes() == want


def test_binary_encoder_paths_with_and_without_room_write_the_same_bytes():
    """fpcc_rans_binary_encode takes its branch-free step when the buffer holds the worst case (2 bytes per symbol) and the
    renormalisation loop otherwise: both must write the oracle's bytes; a buffer that is too small for the stream is an error, a zero
    probability an invalid argument on either path"""
    from fastpcc_amd._native import host
    rng = np.random.default_rng(11)
    for n in (1, 7, 4096, 50001):
        prob = np.clip(np.round(rng.beta(0.4, 0.4, n) * 65536), 1, 65535).astype(np.uint16)
        bits = (rng.random(n) < prob / 65536).astype(np.uint8)
        want = orc.BinaryRansCoder(1).encode(bits.astype(bool)[None], prob.astype(np.uint32)[None])[0]
        for cap in (4 * n + 64, 2 * n + 8, 2 * n + 7, len(want) + 1, len(want)):          # roomy, exactly roomy, loop, loop, loop (tight)
            out = np.empty(cap, np.uint8)
            ln = host().fpcc_rans_binary_encode(bits.ctypes.data, prob.ctypes.data, n, out.ctypes.data, cap)
            assert ln == len(want) and out[cap - ln:].tobytes() == want, (n, cap)
        out = np.empty(max(len(want) - 1, 1), np.uint8)
        assert host().fpcc_rans_binary_encode(bits.ctypes.data, prob.ctypes.data, n, out.ctypes.data, len(want) - 1) < 0
        bad = prob.copy()
        bad[n // 2] = 0
        for cap in (4 * n + 64, 2 * n + 7):
            out = np.empty(cap, np.uint8)
            assert host().fpcc_rans_binary_encode(bits.ctypes.data, bad.ctypes.data, n, out.ctypes.data, cap) < 0


def test_wide_row_decoder_with_row_warmers_and_concurrent_decoders():
    """the 255-ary decoder on blocks of >= 2048 cold rows can start the library's row-warming helpers (FPCC_HOST_WARMERS, read once per
    process: off by default, so this runs in a child process with two helpers; one decoder at a time owns them, a second decoder running
    meanwhile goes without); same symbols either way, from several threads at once"""
    import os
    import subprocess
    import sys
    code = r'''
import threading
import numpy as np
from fastpcc_amd.rans_coder import RansDecoder, RansEncoder
rng = np.random.default_rng(21)
n = 6000
f = rng.integers(1, 300, (n, 255)).astype(np.int64)
f = (f * 65000 // f.sum(1, keepdims=True)) + 1
cdf = np.cumsum(f, 1)
cdf[:, -1] = 65535
rows = cdf.astype(np.uint16)
slot = rng.integers(0, 65535, n).astype(np.uint16)
sym = (rows <= slot[:, None]).sum(1).clip(0, 254).astype(np.uint16)
enc = RansEncoder(8 * 1024 * 1024)
enc.encode(rows, sym)
data = enc.flush()
results = [None] * 4
def work(i):
    dec = RansDecoder()
    dec.flush(data)
    out = np.empty(n, dtype=np.uint16)
    for a in range(0, n, 3000):                       # two blocks of 3000 rows x 510 bytes: above the warmers' threshold
        dec.decode(rows[a:a + 3000], out[a:a + 3000])
    results[i] = out
threads = [threading.Thread(target=work, args=(i,)) for i in range(4)]
for t in threads: t.start()
for t in threads: t.join()
assert all((out == sym).all() for out in results)
print('ok')
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for warmers in ('2', '0'):
        env = dict(os.environ, FPCC_HOST_WARMERS=warmers, PYTHONPATH=root + os.pathsep + os.environ.get('PYTHONPATH', ''))
        out = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True, timeout=120)
        assert out.returncode == 0 and out.stdout.strip().endswith('ok'), out.stderr[-2000:]


def test_resolved_ranges_write_the_bytes_of_row_coding():
    """RansEncoder.encode_ranges (the integer codec's encoder: symbols resolved to (start, freq - 1) on the device) against
    RansEncoder.encode on the CDF rows themselves -- the stream format's definition -- for peaked and flat rows, frequencies 1 and
    65 535, with and without the room for the branch-free step (two spare bytes per symbol) in the coder's buffer"""
    import time
    from fastpcc_amd.rans_coder import RansDecoder, RansEncoder
    rng = np.random.default_rng(11)
    n, width = 6000, 255
    # strictly increasing rows that end at 65 535: a few distinct cuts close together (peaked: most symbols have small frequencies, the
    # gaps are large) or spread over the whole range (flat)
    cdf = np.empty((n, width), dtype=np.int64)
    for i in range(n):
        span = 65534 if i % 3 else 2000
        base = int(rng.integers(1, 65535 - span)) if span < 65534 else 1
        cdf[i, :-1] = base + np.sort(rng.choice(span, width - 1, replace=False))
    cdf[:, -1] = 65535
    cdf[0] = np.where(np.arange(width) < 7, np.arange(width) + 1, 65535 - (width - 1 - np.arange(width)))     # symbol 7 takes nearly all
    rows = cdf.astype(np.uint16)
    sym = rng.integers(0, width, n).astype(np.uint16)
    sym[0] = 7
    lo = np.where(sym == 0, 0, rows[np.arange(n), np.maximum(sym, 1) - 1]).astype(np.int64)
    hi = rows[np.arange(n), sym].astype(np.int64)
    hi[sym == width - 1] = 65536                                        # the last edge stands for 2^16
    assert (hi > lo).all()
    want_enc = RansEncoder(1 << 20)
    want_enc.encode(rows, sym)
    want = want_enc.flush()
    for cap in (1 << 20, len(want) + 64):                        # roomy buffer: branch-free step; tight buffer: the checked loop
        enc = RansEncoder(cap)
        enc.encode_ranges(lo.astype(np.uint16), (hi - lo - 1).astype(np.uint16))
        assert enc.flush() == want
    dec = RansDecoder()
    dec.flush(want)
    back = np.empty(n, dtype=np.uint16)
    dec.decode(rows, back)
    assert (back == sym).all()
