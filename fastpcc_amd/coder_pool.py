"""Background entropy-coding jobs on libfpcc_host's own threads (include/fpcc_host.h, "Background coder pool").

The jobs run the same coders as fastpcc_amd.rans_coder; what they add is WHEN they run: a job fires as soon as a flag in
pinned host memory flips, and the caller orders that flip after the device->host copies of the job's inputs, so coding a
pyramid level overlaps the GPU work of the following levels.  All buffers are NumPy views of caller-owned (pinned) memory
and must stay alive until wait().
"""
from typing import List, Optional, Tuple

import numpy as np

from ._native import host, host_check


class CoderPool:
    def __init__(self, n_threads: int = 4):
        self._n = int(n_threads)
        self._h = host().fpcc_pool_new(self._n)
        if not self._h:
            raise RuntimeError('libfpcc_host: cannot create the coder pool')
        self._keep: List = []
        self._binary: List[Tuple[np.ndarray, np.ndarray]] = []

    def __deepcopy__(self, memo):                 # native threads are not copyable: a copy owns a fresh pool
        return CoderPool(self._n)

    def close(self):
        if self._h:
            host().fpcc_pool_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def _flag(flag: Optional[np.ndarray]):
        if flag is None:
            return None
        if flag.dtype != np.uint32 or flag.size != 1:
            raise ValueError('a flag is one uint32 in host memory')
        return flag.ctypes.data

    def binary_encode(self, bits: np.ndarray, prob1: np.ndarray, flag: Optional[np.ndarray] = None, ready: int = 1) -> int:
        """queue one binary stream; returns its ticket for results()"""
        if bits.dtype not in (np.uint8, np.bool_) or prob1.dtype != np.uint16 or bits.shape != prob1.shape or bits.ndim != 1:
            raise ValueError('bits: uint8/bool [n], prob1: uint16 [n]')
        n = bits.size
        out = np.empty(4 * n + 64, dtype=np.uint8)
        length = np.zeros(1, dtype=np.int64)
        host_check(host().fpcc_pool_binary_encode(self._h, self._flag(flag), ready, bits.ctypes.data, prob1.ctypes.data, n,
                                                  out.ctypes.data, out.size, length.ctypes.data))
        self._keep.append((bits, prob1, flag))
        self._binary.append((out, length))
        return len(self._binary) - 1

    def histogram_encode(self, symbols: np.ndarray, offset: Optional[int] = None, flag: Optional[np.ndarray] = None,
                         ready: int = 1):
        """queue `rans_encode_with_cdf` of an int32 array; returns a handle read by histogram_result() after wait()"""
        if symbols.dtype != np.int32 or not symbols.flags.c_contiguous:
            raise ValueError('symbols: contiguous int32')
        n = symbols.size
        out = np.empty(4 * n + 64, dtype=np.uint8)
        cdf = np.empty(1 << 12, dtype=np.uint32)
        meta = np.zeros(3, dtype=np.int64)             # cdf_len, stream length, (unused)
        off = np.array([0 if offset is None else offset], dtype=np.int32)
        host_check(host().fpcc_pool_histogram_encode(self._h, self._flag(flag), ready, symbols.ctypes.data, n,
                                                     int(offset is not None), off.ctypes.data, cdf.ctypes.data, cdf.size,
                                                     meta[0:].ctypes.data, out.ctypes.data, out.size, meta[1:].ctypes.data))
        self._keep.append((symbols, flag))
        return out, cdf, meta, off

    @staticmethod
    def histogram_result(handle) -> Tuple[int, List[int], bytes]:
        out, cdf, meta, off = handle
        host_check(int(meta[1]))
        return int(off[0]), cdf[:int(meta[0])].tolist(), out[out.size - int(meta[1]):].tobytes()

    def table_decode(self, stream: bytes, n: int, cdf: List[int], offset: int, out: np.ndarray, first_chunk: int = 0):
        """queue the decode of a single-table stream into `out` (int32 [n]); returns a progress handle for need()"""
        if out.dtype != np.int32 or out.size != n or not out.flags.c_contiguous:
            raise ValueError('out: contiguous int32 [n]')
        data = np.frombuffer(stream, dtype=np.uint8)
        table = np.asarray(cdf, dtype=np.uint32)
        progress = np.zeros(1, dtype=np.int64)
        host_check(host().fpcc_pool_table_decode(self._h, data.ctypes.data, data.size, n, table.ctypes.data, table.size,
                                                 int(offset), out.ctypes.data, int(first_chunk), progress.ctypes.data))
        self._keep.append((stream, data, table, out, progress))
        return progress

    def binary_decode(self, stream: np.ndarray, prob1: np.ndarray, out: np.ndarray) -> np.ndarray:
        """queue the decode of one binary stream (uint8 [len]) under prob1 (uint16 [n]) into out (uint8 [n]); returns the job's
        completion word for need(word, 1)"""
        if stream.dtype != np.uint8 or prob1.dtype != np.uint16 or out.dtype != np.uint8 or prob1.shape != out.shape or out.ndim != 1:
            raise ValueError('stream: uint8 [len], prob1: uint16 [n], out: uint8 [n]')
        done = np.zeros(1, dtype=np.int64)
        host_check(host().fpcc_pool_binary_decode(self._h, stream.ctypes.data, stream.size, prob1.ctypes.data, out.size,
                                                  out.ctypes.data, done.ctypes.data))
        self._keep.append((stream, prob1, out, done))
        return done

    @staticmethod
    def need(progress: np.ndarray, count: int) -> None:
        """block until the first `count` symbols of a table_decode are final"""
        host_check(host().fpcc_progress_wait(progress.ctypes.data, int(count)))

    def wait(self) -> List[bytes]:
        """block until every queued job has finished; returns the binary streams in ticket order"""
        rc = host().fpcc_pool_wait(self._h)
        binary, self._binary = self._binary, []
        self._keep = []
        if rc < 0:                                 # which job: the per-job status words say (negative = that job's error)
            self.last_failure = [int(length[0]) for _, length in binary]
        host_check(rc)
        return [out[out.size - int(host_check(int(length[0]))):].tobytes() for out, length in binary]
