"""TEST ORACLE -- not product code.  ctypes face of oracle/rans.c.

Class and method names mirror the reference's pybind modules so that tests read like the reference's self-tests
(/root/reference/lib/entropy_models/rans_coder/__init__.py:9-96,
 /root/reference/models/convolutional/lossy_coord_v3/rans_coder/__init__.py:28-63).
"""
import ctypes as C
from typing import List, Optional, Sequence

import numpy as np

from . import lib

_i64 = C.c_int64
_vp = C.c_void_p


def _p(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(_vp)


def _setup():
    L = lib()
    L.orc_pmf_to_cdf.restype = _i64
    L.orc_pmf_to_cdf.argtypes = [_vp, _i64, C.c_int, _vp, _vp]
    L.orc_indexed_encode.restype = _i64
    L.orc_indexed_encode.argtypes = [_vp, _vp, _i64, _vp, _vp, _vp, _vp, _i64, C.c_int, _vp, _i64]
    L.orc_indexed_decode.restype = None
    L.orc_indexed_decode.argtypes = [_vp, _vp, _i64, _vp, _vp, _vp, _vp, _i64, C.c_int, _vp]
    L.orc_binary_encode.restype = _i64
    L.orc_binary_encode.argtypes = [_vp, _vp, _i64, _vp, _i64]
    L.orc_binary_decode.restype = None
    L.orc_binary_decode.argtypes = [_vp, _vp, _i64, _vp]
    L.orc_simple_enc_new.restype = _vp
    L.orc_simple_enc_new.argtypes = [_i64]
    L.orc_simple_enc_free.argtypes = [_vp]
    L.orc_simple_enc_push.restype = _i64
    L.orc_simple_enc_push.argtypes = [_vp, _vp, _i64, _i64, _vp, _i64]
    L.orc_simple_enc_push_bin.restype = _i64
    L.orc_simple_enc_push_bin.argtypes = [_vp, _vp, _i64, _vp, _i64]
    L.orc_simple_enc_finish.restype = _i64
    L.orc_simple_enc_finish.argtypes = [_vp, _vp, _i64]
    L.orc_simple_dec_new.restype = _vp
    L.orc_simple_dec_new.argtypes = [_vp]
    L.orc_simple_dec_free.argtypes = [_vp]
    L.orc_simple_dec_pop.restype = None
    L.orc_simple_dec_pop.argtypes = [_vp, _vp, _i64, _i64, _vp, _i64]
    L.orc_simple_dec_pop_bin.restype = None
    L.orc_simple_dec_pop_bin.argtypes = [_vp, _vp, _i64, _vp, _i64]
    return L


_L = None


def _lib():
    global _L
    if _L is None:
        _L = _setup()
    return _L


def batched_pmf_to_quantized_cdf(pmf: np.ndarray, offset: np.ndarray, overflow_coding: bool) -> List[List[int]]:
    """Mutates ``offset`` in overflow mode, like the reference (cdf_ops.cpp:51)."""
    assert pmf.ndim == 2 and offset.dtype == np.int32 and offset.shape == (pmf.shape[0],)
    out = []
    for i in range(pmf.shape[0]):
        row = np.array(pmf[i], dtype=np.float64, order='C')
        cdf = np.zeros(row.size + 2, dtype=np.uint32)
        off = np.array([offset[i]], dtype=np.int32)
        n = _lib().orc_pmf_to_cdf(_p(row), row.size, int(overflow_coding), _p(off), _p(cdf))
        assert n > 0, 'no bin can donate a count'
        offset[i] = off[0]
        out.append(cdf[:n].tolist())
    return out


class IndexedRansCoder:
    def __init__(self, overflow_coding: bool, batch_size: int, enc_buf_size: int = 8 << 20):
        self.overflow_coding = bool(overflow_coding)
        self.batch_size = batch_size
        self._cdfs: List[List[int]] = []
        self._offsets = np.zeros(0, np.int32)

    def init_with_pmfs(self, pmf_array: np.ndarray, offset_array: np.ndarray):
        cdfs = batched_pmf_to_quantized_cdf(np.asarray(pmf_array, np.float64), offset_array, self.overflow_coding)
        return self.init_with_quantized_cdfs(cdfs, offset_array)

    def init_with_quantized_cdfs(self, cdfs: Sequence[Sequence[int]], offset_array: np.ndarray):
        self._cdfs = [list(map(int, c)) for c in cdfs]
        for c in self._cdfs:
            assert c[0] == 0 and c[-1] == 1 << 16
        self._offsets = np.array(offset_array, dtype=np.int32)
        self._flat = np.array([v for c in self._cdfs for v in c], dtype=np.uint32)
        lens = np.array([len(c) for c in self._cdfs], dtype=np.int64)
        self._len = lens
        self._start = np.concatenate(([0], np.cumsum(lens)[:-1])).astype(np.int64)
        return 0

    def get_cdfs(self):
        return [list(c) for c in self._cdfs]

    def get_offset_array(self):
        return self._offsets

    def _enc(self, symbols, indexes):
        symbols = np.ascontiguousarray(symbols, dtype=np.int32)
        assert symbols.ndim == 2 and symbols.shape[0] == self.batch_size
        out = []
        for b in range(self.batch_size):
            n = symbols.shape[1]
            cap = 8 * n + 64
            buf = np.empty(cap, np.uint8)
            idx = None if indexes is None else np.ascontiguousarray(indexes[b], dtype=np.int32)
            sym_b = np.ascontiguousarray(symbols[b])
            got = _lib().orc_indexed_encode(_p(sym_b), _p(idx), n, _p(self._flat), _p(self._start),
                                            _p(self._len), _p(self._offsets), len(self._cdfs),
                                            int(self.overflow_coding), _p(buf), cap)
            assert got >= 0
            out.append(buf[cap - got:].tobytes())
        return out

    def encode(self, symbol_array):
        return self._enc(symbol_array, None)

    def encode_with_indexes(self, symbol_array, index_array):
        return self._enc(symbol_array, np.asarray(index_array))

    def _dec(self, encoded_list, indexes, symbol_array):
        assert symbol_array.dtype == np.int32 and symbol_array.flags.c_contiguous
        for b in range(self.batch_size):
            data = np.frombuffer(encoded_list[b] + b'\0' * 8, dtype=np.uint8)
            idx = None if indexes is None else np.ascontiguousarray(indexes[b], dtype=np.int32)
            row = symbol_array[b]
            _lib().orc_indexed_decode(_p(data), _p(idx), row.size, _p(self._flat), _p(self._start), _p(self._len),
                                      _p(self._offsets), len(self._cdfs), int(self.overflow_coding), _p(row))
        return 0

    def decode(self, encoded_list, symbol_array):
        return self._dec(encoded_list, None, symbol_array)

    def decode_with_indexes(self, encoded_list, index_array, symbol_array):
        return self._dec(encoded_list, np.asarray(index_array), symbol_array)


class BinaryRansCoder:
    def __init__(self, batch_size: int, enc_buf_size: int = 8 << 20):
        self.batch_size = batch_size

    def encode(self, symbol_array: np.ndarray, prob_array: np.ndarray):
        assert symbol_array.shape == prob_array.shape and symbol_array.shape[0] == self.batch_size
        out = []
        for b in range(self.batch_size):
            bits = np.ascontiguousarray(symbol_array[b]).astype(np.uint8)
            prob = np.ascontiguousarray(prob_array[b], dtype=np.uint32)
            cap = 4 * bits.size + 64
            buf = np.empty(cap, np.uint8)
            got = _lib().orc_binary_encode(_p(bits), _p(prob), bits.size, _p(buf), cap)
            assert got >= 0
            out.append(buf[cap - got:].tobytes())
        return out

    def decode(self, encoded_list, prob_array: np.ndarray, symbol_array: np.ndarray):
        assert symbol_array.dtype == np.bool_
        for b in range(self.batch_size):
            data = np.frombuffer(encoded_list[b] + b'\0' * 8, dtype=np.uint8)
            prob = np.ascontiguousarray(prob_array[b], dtype=np.uint32)
            bits = np.empty(prob.size, np.uint8)
            _lib().orc_binary_decode(_p(data), _p(prob), prob.size, _p(bits))
            symbol_array[b] = bits.astype(np.bool_)
        return 0


class RansEncoder:
    """Single persistent stream (reference: simple_rans_wrapper.cpp RansEncoder)."""

    def __init__(self, enc_buf_size: int = 32 << 20):
        self._cap = enc_buf_size
        self._h = _lib().orc_simple_enc_new(enc_buf_size)

    def __del__(self):
        if getattr(self, '_h', None):
            _lib().orc_simple_enc_free(self._h)
            self._h = None

    def encode(self, cdf_arr: np.ndarray, symbol_arr: np.ndarray) -> int:
        cdf_arr = np.ascontiguousarray(cdf_arr, dtype=np.uint16)
        symbol_arr = np.ascontiguousarray(symbol_arr, dtype=np.uint16)
        assert cdf_arr.ndim == 2 and cdf_arr.shape[0] in (1, symbol_arr.shape[0])
        got = _lib().orc_simple_enc_push(self._h, _p(cdf_arr), cdf_arr.shape[0], cdf_arr.shape[1], _p(symbol_arr),
                                         symbol_arr.shape[0])
        assert got >= 0
        return got

    encode_with_precomp = encode  # same stream (RansEncPutSymbol == RansEncPut, rans_byte.h:274-296)

    def encode_bin(self, cdf_arr: np.ndarray, symbol_arr: np.ndarray) -> int:
        cdf_arr = np.ascontiguousarray(cdf_arr, dtype=np.uint16).reshape(-1)
        bits = np.ascontiguousarray(symbol_arr).astype(np.uint8)
        got = _lib().orc_simple_enc_push_bin(self._h, _p(cdf_arr), cdf_arr.shape[0], _p(bits), bits.shape[0])
        assert got >= 0
        return got

    def flush(self) -> bytes:
        buf = np.empty(self._cap, np.uint8)
        got = _lib().orc_simple_enc_finish(self._h, _p(buf), self._cap)
        assert got >= 0
        return buf[:got].tobytes()


class RansDecoder:
    def __init__(self):
        self._h = None
        self._keep = None

    def __del__(self):
        if getattr(self, '_h', None):
            _lib().orc_simple_dec_free(self._h)
            self._h = None

    def flush(self, encoded: bytes):
        if self._h:
            _lib().orc_simple_dec_free(self._h)
        self._keep = np.frombuffer(encoded + b'\0' * 8, dtype=np.uint8)
        self._h = _lib().orc_simple_dec_new(_p(self._keep))
        return 0

    def decode(self, cdf_arr: np.ndarray, symbol_arr: np.ndarray):
        cdf_arr = np.ascontiguousarray(cdf_arr, dtype=np.uint16)
        assert symbol_arr.dtype == np.uint16 and symbol_arr.flags.c_contiguous
        _lib().orc_simple_dec_pop(self._h, _p(cdf_arr), cdf_arr.shape[0], cdf_arr.shape[1], _p(symbol_arr),
                                  symbol_arr.shape[0])
        return 0

    decode_with_precomp = decode

    def decode_bin(self, cdf_arr: np.ndarray, symbol_arr: np.ndarray):
        cdf_arr = np.ascontiguousarray(cdf_arr, dtype=np.uint16).reshape(-1)
        bits = np.empty(symbol_arr.shape[0], np.uint8)
        _lib().orc_simple_dec_pop_bin(self._h, _p(cdf_arr), cdf_arr.shape[0], _p(bits), bits.shape[0])
        symbol_arr[...] = bits.astype(np.bool_)
        return 0
