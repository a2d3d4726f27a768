"""Background coder pool of libfpcc_host: same bytes as the inline coders, jobs gated by host-visible flags."""
import threading
import time

import numpy as np
import pytest

from fastpcc_amd.coder_pool import CoderPool
from fastpcc_amd.rans_coder import BinaryRansCoder, IndexedRansCoder


def _binary_case(rng, n):
    p = np.clip(np.round(rng.beta(.4, .4, n) * 65536), 1, 65535).astype(np.uint16)
    bits = (rng.random(n) < p / 65536).astype(np.uint8)
    want = BinaryRansCoder(1).encode(bits[None].astype(bool), p[None].astype(np.uint32))[0]
    return bits, p, want


def test_flag_gated_binary_jobs_match_inline_coder():
    rng = np.random.default_rng(5)
    pool = CoderPool(3)
    cases = [_binary_case(rng, n) for n in (1, 100, 50000, 7, 20000)]
    flags = np.zeros(len(cases), dtype=np.uint32)
    # inputs are only valid once the flag flips: start from garbage, fill in + flip from another thread
    bufs = [(np.full_like(b, 1), np.full_like(p, 77)) for b, p, _ in cases]
    for i, (b, p) in enumerate(bufs):
        assert pool.binary_encode(b, p, flags[i:i + 1], ready=i + 1) == i

    def producer():
        for i in reversed(range(len(cases))):          # out of order on purpose
            time.sleep(0.01)
            bufs[i][0][:] = cases[i][0]
            bufs[i][1][:] = cases[i][1]
            flags[i] = i + 1
    th = threading.Thread(target=producer)
    th.start()
    got = pool.wait()
    th.join()
    assert got == [c[2] for c in cases]
    assert pool.wait() == []                           # reusable, nothing pending
    pool.close()


@pytest.mark.parametrize('fixed', [False, True])
def test_histogram_job_matches_rans_encode_with_cdf(fixed):
    rng = np.random.default_rng(6)
    sym = np.clip(np.round(rng.normal(0, 4, 30000)), -20, 20).astype(np.int32) + (25 if fixed else 0)
    pool = CoderPool(2)
    h = pool.histogram_encode(sym, 0 if fixed else None)
    pool.wait()
    offset, cdf, payload = pool.histogram_result(h)
    ref = IndexedRansCoder(False, 1)
    off = np.array([0 if fixed else sym.min()], dtype=np.int32)
    ref.init_with_pmfs(np.bincount(sym - off[0]).astype(np.float64)[None], off)
    assert offset == int(off[0]) and cdf == ref.get_cdfs()[0]
    assert payload == ref.encode(sym[None])[0]
    # ... and the background decoder returns the symbols, publishing progress on the way
    out = np.full(sym.size, -99, dtype=np.int32)
    prog = pool.table_decode(payload, sym.size, cdf, offset, out, first_chunk=10)
    pool.need(prog, 10)
    assert (out[:10] == sym[:10]).all()
    pool.need(prog, sym.size)
    assert (out == sym).all()
    pool.wait()
    pool.close()


def test_job_errors_surface_in_wait():
    pool = CoderPool(1)
    bits = np.zeros(4, dtype=np.uint8)
    prob = np.array([5, 0, 7, 9], dtype=np.uint16)          # probability 0 is not codable
    pool.binary_encode(bits, prob)
    with pytest.raises(RuntimeError):
        pool.wait()
    pool.binary_encode(bits, np.array([5, 1, 7, 9], dtype=np.uint16))
    assert len(pool.wait()) == 1
    pool.close()
