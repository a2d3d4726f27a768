"""Host entropy coders, same Python surface as the reference's two pybind modules:

    IndexedRansCoder, BinaryRansCoder, batched_pmf_to_quantized_cdf
        -> /root/reference/lib/entropy_models/rans_coder/__init__.py:48-50 (rans_ext_cpp)
    RansEncoder, RansDecoder
        -> /root/reference/models/convolutional/lossy_coord_v3/rans_coder/__init__.py:25-26 (simple_rans_ext_cpp)

backed by libfpcc_host.so through its C ABI (include/fpcc_host.h).  Argument meaning follows the reference; where the
reference aborts through a C assert, these raise RuntimeError/ValueError instead.
"""
from typing import List, Optional, Sequence

import numpy as np

from ._native import host, host_check


def _p(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data


def batched_pmf_to_quantized_cdf(pmf_array: np.ndarray, offset_array: np.ndarray, overflow_coding: bool):
    pmf_array = np.ascontiguousarray(pmf_array, dtype=np.float64)
    if pmf_array.ndim != 2 or offset_array.dtype != np.int32 or offset_array.shape != (pmf_array.shape[0],):
        raise ValueError('pmf must be [B, S] float64 and offset [B] int32')
    out = []
    scratch = np.empty(pmf_array.shape[1] + 2, dtype=np.uint32)
    one = np.empty(1, dtype=np.int32)
    for b in range(pmf_array.shape[0]):
        one[0] = offset_array[b]
        n = host_check(host().fpcc_pmf_to_quantized_cdf(_p(pmf_array[b]), pmf_array.shape[1], int(overflow_coding),
                                                        _p(one), _p(scratch)))
        offset_array[b] = one[0]
        out.append(scratch[:n].tolist())
    return out


class IndexedRansCoder:
    def __init__(self, overflow_coding: bool, batch_size: int, enc_buf_size: int = 8 << 20):
        if batch_size < 1:
            raise ValueError('batch_size must be positive')
        self.overflow_coding = bool(overflow_coding)
        self.batch_size = int(batch_size)
        self._tables: List[List[int]] = []
        self._offsets = np.zeros(0, np.int32)
        self._flat = self._start = self._len = None

    # -- table set-up ---------------------------------------------------------------------------------------------
    def init_with_pmfs(self, pmf_array: np.ndarray, offset_array: np.ndarray) -> int:
        return self.init_with_quantized_cdfs(
            batched_pmf_to_quantized_cdf(pmf_array, offset_array, self.overflow_coding), offset_array)

    def init_with_quantized_cdfs(self, quantized_cdfs: Sequence[Sequence[int]], offset_array: np.ndarray) -> int:
        tables = [[int(v) for v in row] for row in quantized_cdfs]
        for row in tables:
            if len(row) < 2 or row[0] != 0 or row[-1] != 1 << 16:
                raise ValueError('a quantised CDF must start at 0 and end at 65536')
        self._tables = tables
        self._offsets = np.array(offset_array, dtype=np.int32).reshape(-1)
        lens = np.fromiter((len(r) for r in tables), dtype=np.int64, count=len(tables))
        self._len = lens
        self._start = (np.cumsum(lens) - lens).astype(np.int64)
        self._flat = np.fromiter((v for r in tables for v in r), dtype=np.uint32, count=int(lens.sum()))
        return 0

    def get_cdfs(self):
        return [list(r) for r in self._tables]

    def get_offset_array(self):
        return self._offsets

    # -- coding ---------------------------------------------------------------------------------------------------
    def _encode(self, symbol_array, index_array) -> List[bytes]:
        sym = np.ascontiguousarray(symbol_array, dtype=np.int32)
        if sym.ndim != 2 or sym.shape[0] != self.batch_size:
            raise ValueError('symbols must be [batch_size, n] int32')
        idx = None
        if index_array is not None:
            idx = np.ascontiguousarray(index_array, dtype=np.int32)
            if idx.shape != sym.shape:
                raise ValueError('indexes must have the shape of symbols')
        n = sym.shape[1]
        cap = max(64, 4 * n + 64) if not self.overflow_coding else 40 * n + 64
        buf = np.empty(cap, dtype=np.uint8)
        out = []
        for b in range(self.batch_size):
            got = host_check(host().fpcc_rans_indexed_encode(
                _p(sym[b]), None if idx is None else _p(idx[b]), n, _p(self._flat), _p(self._start), _p(self._len),
                _p(self._offsets), len(self._tables), int(self.overflow_coding), _p(buf), cap))
            out.append(buf[cap - got:].tobytes())
        return out

    def encode(self, symbol_array) -> List[bytes]:
        return self._encode(symbol_array, None)

    def encode_with_indexes(self, symbol_array, index_array) -> List[bytes]:
        return self._encode(symbol_array, index_array)

    def _decode(self, encoded_list, index_array, symbol_array) -> int:
        if len(encoded_list) != self.batch_size:
            raise ValueError('one byte string per batch unit expected')
        if symbol_array.dtype != np.int32 or not symbol_array.flags.c_contiguous or not symbol_array.flags.writeable:
            raise ValueError('output must be a writable C-contiguous int32 array')
        idx = None if index_array is None else np.ascontiguousarray(index_array, dtype=np.int32)
        for b in range(self.batch_size):
            data = np.frombuffer(encoded_list[b], dtype=np.uint8)
            row = symbol_array[b]
            host_check(host().fpcc_rans_indexed_decode(
                _p(data), data.size, None if idx is None else _p(idx[b]), row.size, _p(self._flat), _p(self._start),
                _p(self._len), _p(self._offsets), len(self._tables), int(self.overflow_coding), _p(row)))
        return 0

    def decode(self, encoded_list, symbol_array) -> int:
        return self._decode(encoded_list, None, symbol_array)

    def decode_with_indexes(self, encoded_list, index_array, symbol_array) -> int:
        return self._decode(encoded_list, index_array, symbol_array)


class BinaryRansCoder:
    """``prob_array`` holds P(symbol == 1) * 65536 as integers in [1, 65535] (any integer dtype)."""

    def __init__(self, batch_size: int, enc_buf_size: int = 8 << 20):
        if batch_size < 1:
            raise ValueError('batch_size must be positive')
        self.batch_size = int(batch_size)

    def encode(self, symbol_array: np.ndarray, prob_array: np.ndarray) -> List[bytes]:
        if symbol_array.shape != prob_array.shape or symbol_array.ndim != 2 or symbol_array.shape[0] != self.batch_size:
            raise ValueError('symbols and probabilities must both be [batch_size, n]')
        bits = np.ascontiguousarray(symbol_array).view(np.uint8) if symbol_array.dtype == np.bool_ \
            else np.ascontiguousarray(symbol_array != 0).view(np.uint8)
        prob = np.ascontiguousarray(prob_array, dtype=np.uint16)
        n = bits.shape[1]
        cap = 4 * n + 64
        buf = np.empty(cap, dtype=np.uint8)
        out = []
        for b in range(self.batch_size):
            got = host_check(host().fpcc_rans_binary_encode(_p(bits[b]), _p(prob[b]), n, _p(buf), cap))
            out.append(buf[cap - got:].tobytes())
        return out

    def decode(self, encoded_list, prob_array: np.ndarray, symbol_array: np.ndarray) -> int:
        if symbol_array.dtype != np.bool_ or not symbol_array.flags.c_contiguous:
            raise ValueError('output must be a C-contiguous bool array')
        prob = np.ascontiguousarray(prob_array, dtype=np.uint16)
        for b in range(self.batch_size):
            data = np.frombuffer(encoded_list[b], dtype=np.uint8)
            row = symbol_array[b].view(np.uint8)
            host_check(host().fpcc_rans_binary_decode(_p(data), data.size, _p(prob[b]), row.size, _p(row)))
        return 0


class RansEncoder:
    def __init__(self, enc_buf_size: int = 32 << 20):
        self._cap = int(enc_buf_size)
        self._h = host().fpcc_simple_enc_new(self._cap)
        if not self._h:
            raise MemoryError('fpcc_simple_enc_new failed')

    def __del__(self):
        h, self._h = getattr(self, '_h', None), None
        if h:
            try:
                host().fpcc_simple_enc_free(h)
            except Exception:                     # interpreter shutdown: the module globals are already gone, the OS takes the memory
                pass

    def __deepcopy__(self, memo):                 # the native handle is not shareable: a copy gets its own (empty) coder
        return RansEncoder(self._cap)

    __copy__ = lambda self: self.__deepcopy__({})

    def encode(self, cdf_arr: np.ndarray, symbol_arr: np.ndarray) -> int:
        rows = np.ascontiguousarray(cdf_arr, dtype=np.uint16)
        sym = np.ascontiguousarray(symbol_arr, dtype=np.uint16)
        if rows.ndim != 2:
            raise ValueError('cdf rows must be 2-D')
        return host_check(host().fpcc_simple_enc_push(self._h, _p(rows), rows.shape[0], rows.shape[1], _p(sym),
                                                      sym.shape[0]))

    encode_with_precomp = encode

    def encode_bin(self, cdf_arr: np.ndarray, symbol_arr: np.ndarray) -> int:
        edge = np.ascontiguousarray(cdf_arr, dtype=np.uint16).reshape(-1)
        bits = np.ascontiguousarray(symbol_arr != 0).view(np.uint8)
        return host_check(host().fpcc_simple_enc_push_bin(self._h, _p(edge), edge.shape[0], _p(bits), bits.shape[0]))

    def encode_ranges(self, start: np.ndarray, freq_minus_1: np.ndarray) -> int:
        """Symbols already resolved to (start, freq-1) uint16 pairs on the device (4 B per symbol over PCIe)."""
        s = np.ascontiguousarray(start, dtype=np.uint16)
        f = np.ascontiguousarray(freq_minus_1, dtype=np.uint16)
        return host_check(host().fpcc_simple_enc_push_ranges(self._h, _p(s), _p(f), s.shape[0]))

    def flush(self) -> bytes:
        buf = np.empty(self._cap, dtype=np.uint8)
        got = host_check(host().fpcc_simple_enc_finish(self._h, _p(buf), self._cap))
        return buf[:got].tobytes()


class RansDecoder:
    def __init__(self):
        self._h = None
        self._pin = None

    def __del__(self):
        h, self._h = getattr(self, '_h', None), None
        if h:
            try:
                host().fpcc_simple_dec_free(h)
            except Exception:
                pass

    def __deepcopy__(self, memo):
        return RansDecoder()

    __copy__ = lambda self: self.__deepcopy__({})

    def flush(self, encoded: bytes) -> int:
        if self._h:
            host().fpcc_simple_dec_free(self._h)
        self._pin = np.frombuffer(encoded, dtype=np.uint8)   # keeps the bytes alive for the decoder
        self._h = host().fpcc_simple_dec_new(_p(self._pin), self._pin.size)
        if not self._h:
            raise ValueError('not a rANS stream: shorter than 4 bytes or a final state below 2^23')
        return 0

    def decode(self, cdf_arr: np.ndarray, symbol_arr: np.ndarray) -> int:
        rows = np.ascontiguousarray(cdf_arr, dtype=np.uint16)
        if symbol_arr.dtype != np.uint16 or not symbol_arr.flags.c_contiguous:
            raise ValueError('output must be C-contiguous uint16')
        return host_check(host().fpcc_simple_dec_pop(self._h, _p(rows), rows.shape[0], rows.shape[1], _p(symbol_arr),
                                                     symbol_arr.shape[0]))

    decode_with_precomp = decode

    def tell(self):
        """(32-bit rANS state, offset of the next unread byte): where a device decoder picks the stream up"""
        state = np.zeros(1, dtype=np.uint32)
        pos = np.zeros(1, dtype=np.int64)
        host_check(host().fpcc_simple_dec_tell(self._h, _p(state), _p(pos)))
        return int(state[0]), int(pos[0])

    def decode_bin(self, cdf_arr: np.ndarray, symbol_arr: np.ndarray) -> int:
        edge = np.ascontiguousarray(cdf_arr, dtype=np.uint16).reshape(-1)
        bits = np.empty(symbol_arr.shape[0], dtype=np.uint8)
        host_check(host().fpcc_simple_dec_pop_bin(self._h, _p(edge), edge.shape[0], _p(bits), bits.shape[0]))
        symbol_arr[...] = bits.view(np.bool_)
        return 0
