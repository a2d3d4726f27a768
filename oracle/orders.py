"""Which fp32 summation order a layer is evaluated in -- the rule of numerics version 3 restated from its specification
(include/fpcc_hip.h, "Numerics version"), independently of the product's implementation (fastpcc_amd/engine.py:summation_order,
csrc/hip/conv.hip:fpcc_conv_f32_order_ex).  TEST INFRASTRUCTURE ONLY, like everything under oracle/: tests/test_orders.py holds the
product's rule against this one over every layer shape of the in-scope configurations and a grid around the thresholds, so that a
change of the rule in the product alone -- which would silently regenerate the chain-order fixtures together with the product
(tests/golden/make_golden.py takes the order from here, not from the product) -- fails a CPU test.

    order 0  natural FMA chain                              shapes outside the MFMA path
    order 1  MFMA chain (0,4,1,5,2,6,3,7 in groups of 8)     MFMA shapes with one kernel offset / 8 groups / 16-channel chunks
    order 2  per-offset chains, then the offsets' sum       3x3x3 to ONE output channel with C_in % 16 == 0
    order 3  grouped: four offset groups, ((g0+g1)+g2)+g3   8 <= K <= 27 offsets, one group, C_in and c1 multiples of 32, C_out in {32,64,128}
plus zero-padding of per-point / 3x3x3 shapes to an MFMA shape on maps (of ONE cloud) of at least PAD_MIN_ROWS rows.
"""
NUMERICS_VERSION = 3
PAD_MIN_ROWS = 8192
_MFMA_COLUMNS = (32, 64, 128)


def _mfma_chunk(c1: int, c2: int, c_out: int) -> int:
    """0: not an MFMA shape; else the channel chunk (32 or 16) the chain advances by"""
    if c_out not in _MFMA_COLUMNS:
        return 0
    c_in = c1 + c2
    if c_in % 32 == 0 and c1 % 32 == 0:
        return 32
    if c_in % 16 == 0 and c1 % 16 == 0:
        return 16
    return 0


def _shape_order(c1: int, c2: int, c_out: int, n_offsets: int, groups: int) -> int:
    chunk = _mfma_chunk(c1, c2, c_out)
    if chunk == 32 and 8 <= n_offsets <= 27 and groups == 1:
        return 3
    return 1 if chunk else 0


def _padded(c1: int, c2: int, c_out: int, n_out: int):
    """the MFMA shape a narrow per-point / 3x3x3 layer is zero-padded to on a map of >= PAD_MIN_ROWS rows, or None"""
    if _shape_order(c1, c2, c_out, 1, 1) != 0 or c_out > 128 or c1 + c2 < 4 or (c1 + c2) * c_out < 32 or n_out < PAD_MIN_ROWS:
        return None
    up16 = lambda c: (c + 15) // 16 * 16
    return up16(c1), (up16(c2) if c2 else 0), (32 if c_out <= 32 else 64 if c_out <= 64 else 128)


def summation_order(kind: str, c1: int, c2: int, c_out: int, n_out: int = 0) -> int:
    """kind: 'k1' (per-point) | 'k3' (3x3x3 on one map) | 'k2s2' (2x2x2 stride 2) | 'k2s2T' (its transpose) | 'gen' (generative
    transpose) | 'mlp'; c1 + c2 input channels (two concatenated sources), n_out = output rows of the layer's map"""
    if kind == 'gen' and c2 == 0 and c1 % 16 == 0 and 8 * c_out in (32, 64, 128, 256):
        return 1                                    # the eight octant kernels side by side: one dense GEMM of 8 c_out columns
    if kind == 'k3' and c_out == 1 and c2 == 0 and c1 % 16 == 0:
        return 2
    n_offsets = {'k3': 27, 'k2s2': 8}.get(kind, 1)
    groups = 8 if kind in ('gen', 'k2s2T') else 1
    if kind in ('k1', 'k3'):
        p = _padded(c1, c2, c_out, n_out)
        if p is not None:
            return _shape_order(p[0], p[1], p[2], n_offsets, groups)
    return _shape_order(c1, c2, c_out, n_offsets, groups)
