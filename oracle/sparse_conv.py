"""TEST ORACLE -- not product code.

Float sparse convolution as MinkowskiEngine's CPU backend structures it (per kernel offset: gather the input rows,
one GEMM, scatter-add into the output rows), out = sum_k X[in_k] @ W[k] + b in fp32 -- semantics (v) of SURVEY.md
section 8a.  Two evaluations of the same sum:

    conv_mm      torch index_select -> mm -> index_add_  (the reference-shaped CPU path; also the timed `cpu_baseline`)
    conv_chain   oracle/sparse_conv.c: one fused-multiply-add chain per output element in a stated order, so that a
                 device kernel documenting the same order can be compared BIT FOR BIT

Parity: UNPINNED against MinkowskiEngine (un-vendored, not installable here).  The two evaluations are checked against
each other in tests/test_oracle_float.py::test_conv_mm_equals_chain.
"""
import ctypes as C
from typing import Optional

import numpy as np
import torch

from . import lib
from .coords import KernelMap, dense_table

ACT_NONE, ACT_PRELU, ACT_RELU = 0, 1, 2


def _act(x: torch.Tensor, act: int, slope: float, clip: float) -> torch.Tensor:
    if act == ACT_PRELU:
        x = torch.where(x < 0, x * slope, x)
    elif act == ACT_RELU:
        x = torch.relu(x)
    if clip > 0:
        x = x.clamp(-clip, clip)
    return x


def conv_mm(x: torch.Tensor, kmap: KernelMap, w: torch.Tensor, bias: Optional[torch.Tensor], n_out: int,
            act: int = ACT_NONE, slope: float = 0.0, clip: float = 0.0) -> torch.Tensor:
    """x [n_in, C_in] fp32, w [K, C_in, C_out], bias [C_out] or None."""
    out = torch.zeros((n_out, w.shape[-1]), dtype=torch.float32)
    w = w.reshape(len(kmap), x.shape[1], -1)
    for k, (rows_in, rows_out) in enumerate(kmap):
        if len(rows_in) == 0:
            continue
        out.index_add_(0, torch.from_numpy(rows_out.astype(np.int64)),
                       x.index_select(0, torch.from_numpy(rows_in.astype(np.int64))) @ w[k])
    if bias is not None:
        out = out + bias.reshape(1, -1)
    return _act(out, act, slope, clip)


def _fp(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def conv_chain(x: np.ndarray, table: Optional[np.ndarray], w: np.ndarray, bias: Optional[np.ndarray], n_out: int,
               x2: Optional[np.ndarray] = None, out_map: Optional[np.ndarray] = None, out_rows: Optional[int] = None,
               act: int = ACT_NONE, slope: float = 0.0, clip: float = 0.0, order: int = 0) -> np.ndarray:
    """table [K, n_out] int32 (-1 absent) or None for the identity map (K = 1); see oracle/sparse_conv.c."""
    L = lib()
    fn = L.orc_gather_conv_f32
    fn.restype = None
    fn.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_int64,
                   C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_float, C.c_float,
                   C.c_int]
    x = np.ascontiguousarray(x, dtype=np.float32)
    c1 = x.shape[1]
    c2 = 0
    if x2 is not None:
        x2 = np.ascontiguousarray(x2, dtype=np.float32)
        c2 = x2.shape[1]
    K = 1 if table is None else table.shape[0]
    w = np.ascontiguousarray(w, dtype=np.float32).reshape(K, c1 + c2, -1)
    c_out = w.shape[2]
    if table is not None:
        table = np.ascontiguousarray(table, dtype=np.int32)
        assert table.shape[1] == n_out
    if bias is not None:
        bias = np.ascontiguousarray(bias, dtype=np.float32).reshape(-1)
    if out_map is not None:
        out_map = np.ascontiguousarray(out_map, dtype=np.int32)
    rows = out_rows if out_rows is not None else n_out
    out = np.zeros((rows, c_out), dtype=np.float32)
    fn(_fp(x), c1, c1, _fp(x2), c2, c2, _fp(table), K, n_out, _fp(w), _fp(bias), c_out, _fp(out_map), _fp(out), c_out,
       act, slope, clip, order)
    return out


def conv_chain_kmap(x: np.ndarray, kmap: KernelMap, w, bias, n_out: int, **kw) -> np.ndarray:
    return conv_chain(x, dense_table(kmap, n_out), w, bias, n_out, **kw)


def sigmoid_spec(x) -> np.ndarray:
    """the logistic function as the HIP path specifies it from numerics version 3 on (oracle/sparse_conv.c:sigmoid_spec): bit for bit
    what fpcc_logit_to_prob16 evaluates on the device"""
    a = np.ascontiguousarray(x.detach().cpu().numpy() if isinstance(x, torch.Tensor) else x, dtype=np.float32)
    out = np.empty_like(a)
    fn = lib().orc_sigmoid_spec_f32
    fn.restype = None
    fn.argtypes = [C.c_void_p, C.c_int64, C.c_void_p]
    fn(a.ctypes.data_as(C.c_void_p), a.size, out.ctypes.data_as(C.c_void_p))
    return out
