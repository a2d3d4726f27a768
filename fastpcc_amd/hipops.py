"""Thin typed wrappers over the C ABI of libfpcc_hip.so (include/fpcc_hip.h).

Tensors are torch CUDA(=HIP) tensors used purely as device buffers: every wrapper checks device / dtype / contiguity,
passes raw pointers plus the current HIP stream, and raises on a non-zero status.  No computation happens in Python and
there is no alternative implementation: without the library or a GPU these functions raise.
"""
import collections
import ctypes as C
import functools
from typing import List, Optional, Tuple

import torch

from . import _native

_vp, _i64, _i32, _f32 = C.c_void_p, C.c_int64, C.c_int, C.c_float

ACT_NONE, ACT_PRELU, ACT_RELU = 0, 1, 2

_SIGS = {
    # name: (restype, argtypes)
    'fpcc_morton3d_encode': (_i32, [_vp, _i64, _i64, _i32, _i32, _i32, _vp, _vp]),
    'fpcc_hilbert3d_encode': (_i32, [_vp, _i64, _i64, _i32, _i32, _i32, _i32, _vp, _vp]),
    'fpcc_keys_from_coords': (_i32, [_vp, _i64, _i32, _i32, _vp, _vp]),
    'fpcc_coords_from_keys': (_i32, [_vp, _i64, _i32, _i32, _vp, _vp, _vp]),
    'fpcc_sort_keys': (_i64, [_vp, _i64, _i32, _vp, _vp, _vp, _i64, _vp]),
    'fpcc_unique_keys': (_i64, [_vp, _i64, _vp, _vp, _vp, _vp, _i64, _vp]),
    'fpcc_coarsen': (_i64, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    'fpcc_octree_level': (_i64, [_vp, _i64, _i64, _i32, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _i64, _vp]),
    'fpcc_level_histogram': (_i64, [_vp, _i64, _i32, _vp, _vp]),
    'fpcc_level_histogram_clouds': (_i64, [_vp, _i64, _i32, _i32, _i32, _vp, _vp]),
    'fpcc_refine': (_i64, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    'fpcc_nbr27_search': (_i32, [_vp, _i64, _i32, _vp, _vp]),
    'fpcc_nbr27_from_parent': (_i32, [_vp, _vp, _i64, _vp, _i64, _vp, _vp, _vp]),
    'fpcc_nbr27_from_parent_ex': (_i32, [_vp, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _vp]),
    'fpcc_conv_row_keys_masks': (_i32, [_vp, _i64, _i32, _vp, _vp]),
    'fpcc_gather_table_rows_i32': (_i32, [_vp, _i32, _vp, _i64, _vp, _vp]),
    'fpcc_mask27_from_parent': (_i32, [_vp, _vp, _i64, _vp, _i64, _vp, _vp, _vp]),
    'fpcc_conv_ones_k3_f32': (_i32, [_vp, _i64, _vp, _vp, _i32, _i32, _vp, C.c_float, _vp, _i32, _vp]),
    'fpcc_conv_f32': (_i32, [_vp, _i32, _i32, _vp, _i32, _i32, _vp, _i32, _i64, _i64, _vp, _vp, _i32, _i32,
                             _vp, _i64, _i64, _vp, _i32, _i64, _i32, _vp, _f32, _vp, _vp, _i64, _vp]),
    'fpcc_conv_f32_pk': (_i32, [_vp, _i32, _i32, _vp, _i32, _i32, _vp, _i32, _i64, _i64, _vp, _vp, _vp, _i32, _i32,
                                _vp, _i64, _i64, _vp, _i32, _i64, _i32, _vp, _f32, _vp, _vp, _i64, _vp]),
    'fpcc_conv_set_tuning': (_i32, [_i32, _i32]),
    'fpcc_conv_debug_stamps': (_i32, [_vp, _i64]),
    'fpcc_conv_i8_debug_stamps': (_i32, [_vp, _i64]),
    'fpcc_time_next_launch': (_i32, [_vp, _vp]),
    'fpcc_transpose_table_i32': (_i32, [_vp, _i32, _i64, _vp, _i32, _vp]),
    'fpcc_numerics_version': (_i32, []),
    'fpcc_conv_packed_floats': (_i64, [_i32, _i32, _i32, _i32, _i32]),
    'fpcc_conv_pack_weights_f32': (_i32, [_vp, _i64, _i32, _i32, _vp, _vp]),
    'fpcc_nn_dist2': (_i32, [_vp, _i64, _i32, _vp, _i64, _vp, _vp, _vp]),
    'fpcc_knn3d': (_i32, [_vp, _i64, _vp, _i64, _i32, _vp, _vp, _vp]),
    'fpcc_sum_i64': (_i32, [_vp, _i64, _vp, _vp]),
    'fpcc_knn_voxels': (_i32, [_vp, _i64, _i32, _vp, _i64, _i32, _i32, _vp, _vp, _vp]),
    'fpcc_pca_normals': (_i32, [_vp, _i64, _i32, _vp, _i64, _i32, _vp, _vp]),
    'fpcc_nn_plane_dist2': (_i32, [_vp, _i64, _i32, _vp, _vp, _i64, _vp, _vp, _vp, _vp]),
    'fpcc_transfer_normals_ws_bytes': (_i64, [_i64, _i64]),
    'fpcc_transfer_normals': (_i32, [_vp, _i64, _vp, _vp, _vp, _i64, _vp, _i32, _vp, _vp, _i64, _vp]),
    'fpcc_sum_max_f64': (_i32, [_vp, _i64, _vp, _vp, _i64, _vp]),
    'fpcc_transpose_weights_f32': (_i32, [_vp, _i32, _i32, _i32, _i32, _vp, _vp]),
    'fpcc_conv_wgrad_ws_bytes': (_i64, [_i32, _i32, _i32, _i32, _i64]),
    'fpcc_conv_wgrad_f32': (_i32, [_vp, _i32, _i32, _vp, _i32, _i32, _vp, _i32, _i64, _i64, _vp, _i64, _i64, _i32, _i64, _vp, _vp,
                                   _i32, _vp, _i64, _vp]),
    'fpcc_epilogue_bwd_ws_bytes': (_i64, [_i64, _i32]),
    'fpcc_epilogue_bwd_f32': (_i32, [_vp, _i32, _vp, _i32, _i64, _i32, _i32, _vp, _vp, _i32, _vp, _vp, _vp, _i64, _vp]),
    'fpcc_noisy_normal_ws_bytes': (_i64, [_i64]),
    'fpcc_noisy_normal_bits_f32': (_i32, [_vp, _vp, _i64, _f32, _f32, _f32, _vp, _vp, _vp, _vp, _i64, _vp]),
    'fpcc_deep_factorized_ws_bytes': (_i64, [_i64, _i32]),
    'fpcc_deep_factorized_bits_f32': (_i32, [_vp, _i64, _i32, _i32, _vp, _vp, _vp, _f32, _vp, _i32, _vp, _vp, _i64, _vp]),
    'fpcc_conv_tile_keys': (_i32, [_vp, _i32, _vp, _i64, _i32, _vp, _vp]),
    'fpcc_conv_regroup_rows': (_i32, [_vp, _vp, _i64, _i32, _vp, _vp]),
    'fpcc_conv_group_order': (_i32, [_vp, _i64, _vp, _vp]),
    'fpcc_conv_row_keys': (_i32, [_vp, _i32, _i64, _i64, _i64, _i32, _vp, _vp, _vp]),
    'fpcc_conv_f32_ws_bytes': (_i64, [_i32, _i32, _i32, _i32, _i32, _i64]),
    'fpcc_conv_f32_order': (_i32, [_i32, _i32, _i32]),
    'fpcc_conv_f32_order_ex': (_i32, [_i32, _i32, _i32, _i32, _i32, _i64]),
    'fpcc_gather_sum_f32': (_i32, [_vp, _i32, _vp, _i32, _i64, _i64, _i64, _vp, _i32, _vp, _f32, _vp, _vp]),
    'fpcc_gather_sum_generated_f32': (_i32, [_vp, _i32, _vp, _i64, _vp, _i32, _vp, _f32, _vp, _vp]),
    'fpcc_logit_to_prob16': (_i32, [_vp, _i64, _vp, _vp]),
    'fpcc_quantize_symbols': (_i32, [_vp, _i64, _f32, _vp, _vp]),
    'fpcc_child_mask': (_i32, [_vp, _i64, _vp, _vp]),
    'fpcc_topk_keep': (_i64, [_vp, _i64, _i64, _vp, _vp, _i64, _vp]),
    'fpcc_topk_keep_cells': (_i64, [_vp, _i64, _vp, _i64, _i64, _vp, _vp, _i64, _vp]),
    'fpcc_gather_rows_f32': (_i32, [_vp, _i32, _i32, _vp, _i64, _vp, _i32, _vp]),
    'fpcc_compact_coords': (_i64, [_vp, _i64, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _i64, _vp]),
    'fpcc_int_init': (_i32, []),
    'fpcc_hash_insert_coords': (_i32, [_vp, _vp, _i64, _vp, _i64, _vp]),
    'fpcc_hash_lookup_coords': (_i32, [_vp, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp]),
    'fpcc_hash_insert_coords_bxyz': (_i32, [_vp, _vp, _i64, _vp, _i64, _vp]),
    'fpcc_hash_lookup_coords_bxyz': (_i32, [_vp, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp]),
    'fpcc_hash_insert_keys': (_i32, [_vp, _vp, _i64, _vp, _i64, _vp]),
    'fpcc_hash_lookup_keys': (_i32, [_vp, _vp, _i64, _vp, _i64, _vp, _vp]),
    'fpcc_conv_i8': (_i32, [_vp, _i32, _i32, _vp, _i32, _i64, _i64, _i32, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _i32, _i32,
                            _vp, _i32, _i32, _i32, _i64, _vp, _vp, _i64, _vp]),
    'fpcc_conv_i8_res': (_i32, [_vp, _i32, _i32, _vp, _i32, _i64, _i64, _i32, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _i32, _i32,
                                _vp, _i32, _i32, _i32, _i64, _vp, _vp, _i32, _vp, _vp, _i64, _vp]),
    'fpcc_conv_i8_ws_bytes': (_i64, [_i32, _i32, _i32, _i32, _i64]),
    'fpcc_epilogue_i32': (_i32, [_vp, _i32, _vp, _vp, _vp, _i32, _vp, _i32, _i32, _vp, _i32, _i32, _i64, _i32, _vp, _vp]),
    'fpcc_prelu_i32': (_i32, [_vp, _vp, _vp, _i64, _vp, _vp]),
    'fpcc_softmax_i32': (_i32, [_vp, _i64, _i32, _vp, _vp]),
    'fpcc_logits_to_cdf16': (_i32, [_vp, _i64, _i32, _i32, _vp, _vp]),
    'fpcc_logits_to_ranges': (_i32, [_vp, _i64, _i32, _i32, _vp, _vp, _vp, _vp]),
    'fpcc_rans_binary_decode_dev': (_i32, [_vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp]),
    'fpcc_simple_dec_pop_dev': (_i32, [_vp, _vp, _i64, _vp, _i64, _i64, _vp, _i64, _vp, _vp]),
    'fpcc_device_count': (_i32, []),
    'fpcc_clock_probe': (_i32, [_vp, _i32, _vp]),
    'fpcc_mlp_chain_f32': (_i32, [_vp, _vp]),
    'fpcc_mlp_chain_set_form': (_i32, [_i32]),
    'fpcc_pointwise_head_f32': (_i32, [_vp, _i32, _i32, _vp, _vp, _i32, _i32, _vp, _i32, _vp, _vp, _i32, _vp, _f32, _vp, _i64, _vp]),
    'fpcc_conv_i8_also': (_i32, [_vp, _i32, _i32, _vp, _i32, _i64, _i64, _i32, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _i32, _i32,
                                 _vp, _i32, _i32, _i32, _i64, _vp, _vp, _i32, _vp, _vp, _i32, _vp, _i64, _vp]),
    'fpcc_epilogue_i32_also': (_i32, [_vp, _i32, _vp, _vp, _vp, _i32, _vp, _i32, _i32, _vp, _i32, _i32, _i64, _i32, _vp, _vp, _i32, _vp]),
    'fpcc_fill_bits_i8': (_i32, [_vp, _i64, _i32, _vp, _vp, _i32, _vp, _i32, _i32, _i32, _vp]),
    'fpcc_int_level_trunk': (_i64, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _i64, _vp]),
    'fpcc_int_level_expand': (_i64, [_vp, _i64, _i64, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    'fpcc_octree_children': (_i64, [_vp, _vp, _vp, _i64, _i64, _i32, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _i64, _vp]),
}
HIP_SYMBOLS = tuple(_SIGS) + ('fpcc_last_error',)
_BLOCKING_ENTRY_POINTS = ('fpcc_hilbert3d_encode', 'fpcc_conv_debug_stamps', 'fpcc_conv_i8_debug_stamps', 'fpcc_int_init')

_lib = None


def lib():
    global _lib
    if _lib is None:
        L = _native.hip()
        blocking = _native.hip_blocking()
        for name, (res, args) in _SIGS.items():
            fn = getattr(L, name)
            if name in _BLOCKING_ENTRY_POINTS:
                # these wait for the device (a table upload followed by a stream synchronise, hipMemcpyToSymbol): bound through the
                # handle that releases the interpreter lock, so another frame's thread keeps enqueueing meanwhile
                fn = getattr(blocking, name)
                setattr(L, name, fn)
            fn.restype = res
            fn.argtypes = args
        L.fpcc_last_error.restype = C.c_char_p
        L.fpcc_last_error.argtypes = []
        _lib = L
    return _lib


class FpccError(RuntimeError):
    pass


def no_gc_pause(fn):
    """Decorator for a codec's compress / decompress: Python's cyclic garbage collector is held off for the duration of the call.
    A frame allocates a few thousand short-lived objects; a generation-2 collection that triggers in the middle of a frame
    stops the host for ~1 ms at whichever launch it happens to precede -- always the same one, since the allocation count per
    frame is constant (measured: one 0.7-ms convolution of the cfg#2 step shows as 1.7 ms in every other run).  The coordinate
    manager's tables are freed by reference counting (engine.clear_global_coordinate_manager cuts its cycles), so nothing here
    depends on the collector."""
    @functools.wraps(fn)
    def wrapped(*args, **kwargs):
        _gc_hold()
        try:
            return fn(*args, **kwargs)
        finally:
            _gc_release()
    return wrapped


# nesting count over all threads: with several frames in flight (fastpcc_amd/serving.py) the collector comes back on only when no
# frame is being coded
_gc_lock = __import__('threading').Lock()
_gc_depth = 0
_gc_was_enabled = False


def _gc_hold():
    import gc
    global _gc_depth, _gc_was_enabled
    with _gc_lock:
        if _gc_depth == 0:
            _gc_was_enabled = gc.isenabled()
            gc.disable()
        _gc_depth += 1


def _gc_release():
    import gc
    global _gc_depth
    with _gc_lock:
        _gc_depth -= 1
        if _gc_depth == 0 and _gc_was_enabled:
            gc.enable()


def _ok(code: int) -> int:
    if code < 0:
        raise FpccError(f'libfpcc_hip status {code}: {lib().fpcc_last_error().decode()}')
    return code


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)
_current_device = getattr(torch._C, '_cuda_getDevice', None)


def _stream() -> int:
    """raw hipStream_t of torch's current stream on the current device (the C entry points of torch: the Python-level
    torch.cuda.current_stream() costs ~9 us a call, more than the launch it precedes)"""
    if _raw_stream is not None and _current_device is not None:
        return _raw_stream(_current_device())
    return torch.cuda.current_stream().cuda_stream


def _dev(t: Optional[torch.Tensor], dtype, name: str, allow_none=False):
    if t is None:
        if allow_none:
            return None
        raise ValueError(f'{name} is required')
    if not t.is_cuda:
        raise FpccError(f'{name} must live on the GPU (libfpcc_hip has no CPU path)')
    if t.dtype != dtype:
        raise TypeError(f'{name} must be {dtype}, got {t.dtype}')
    if not t.is_contiguous():
        raise ValueError(f'{name} must be contiguous')
    return t.data_ptr()


def _ws(query_fn, device) -> Tuple[torch.Tensor, int]:
    need = _ok(query_fn())
    buf = torch.empty(max(int(need), 16), dtype=torch.uint8, device=device)
    return buf, int(need)


# ---------------------------------------------------------------------------------------------------------------
# coordinates

def morton3d_encode(coords: torch.Tensor, cols=(0, 1, 2)) -> torch.Tensor:
    """coords [n, w] int32 (any row stride, unit column stride); cols = columns landing on Morton bits 0, 1, 2."""
    if coords.dtype != torch.int32 or coords.dim() != 2 or not coords.is_cuda:
        raise TypeError('coords must be a 2-D int32 GPU tensor')
    if coords.stride(1) != 1:
        coords = coords.contiguous()
    n = coords.shape[0]
    out = torch.empty(n, dtype=torch.int64, device=coords.device)
    _ok(lib().fpcc_morton3d_encode(coords.data_ptr(), n, coords.stride(0) if n else coords.shape[1], cols[0], cols[1],
                                   cols[2], out.data_ptr(), _stream()))
    return out


def hilbert3d_encode(coords: torch.Tensor, bits: int, cols=(0, 1, 2)) -> torch.Tensor:
    """Hilbert keys (the reference's hilbert3d_encode_lut); cols = the columns that play x, y, z"""
    if coords.dtype != torch.int32 or coords.dim() != 2 or not coords.is_cuda:
        raise TypeError('coords must be a 2-D int32 GPU tensor')
    if coords.stride(1) != 1:
        coords = coords.contiguous()
    n = coords.shape[0]
    out = torch.empty(n, dtype=torch.int64, device=coords.device)
    _ok(lib().fpcc_hilbert3d_encode(coords.data_ptr(), n, coords.stride(0) if n else coords.shape[1], cols[0], cols[1], cols[2],
                                    int(bits), out.data_ptr(), _stream()))
    return out


def keys_from_coords(coords: torch.Tensor, level: int, bits: int) -> torch.Tensor:
    n = coords.shape[0]
    out = torch.empty(n, dtype=torch.int64, device=coords.device)
    _ok(lib().fpcc_keys_from_coords(_dev(coords, torch.int32, 'coords'), n, level, bits, out.data_ptr(), _stream()))
    return out


def coords_from_keys(keys: torch.Tensor, level: int, bits: int, offset_xyz: Optional[torch.Tensor] = None) -> torch.Tensor:
    n = keys.shape[0]
    out = torch.empty((n, 4), dtype=torch.int32, device=keys.device)
    _ok(lib().fpcc_coords_from_keys(_dev(keys, torch.int64, 'keys'), n, level, bits,
                                    _dev(offset_xyz, torch.int32, 'offset', True), out.data_ptr(), _stream()))
    return out


def sort_keys(keys: torch.Tensor, end_bit: int = 63) -> Tuple[torch.Tensor, torch.Tensor]:
    n = keys.shape[0]
    kp = _dev(keys, torch.int64, 'keys')
    out = torch.empty_like(keys)
    perm = torch.empty(n, dtype=torch.int32, device=keys.device)
    L = lib()
    ws, need = _ws(lambda: L.fpcc_sort_keys(None, n, end_bit, None, None, None, 0, None), keys.device)
    _ok(L.fpcc_sort_keys(kp, n, end_bit, out.data_ptr(), perm.data_ptr(), ws.data_ptr(), need, _stream()))
    return out, perm


def unique_keys(keys: torch.Tensor):
    """-> (ukeys[n] (first `count` valid), first[n], count device int32[1])"""
    n = keys.shape[0]
    kp = _dev(keys, torch.int64, 'keys')
    ukeys = torch.empty_like(keys)
    first = torch.empty(n, dtype=torch.int32, device=keys.device)
    count = torch.empty(1, dtype=torch.int32, device=keys.device)
    L = lib()
    ws, need = _ws(lambda: L.fpcc_unique_keys(None, n, None, None, None, None, 0, None), keys.device)
    _ok(L.fpcc_unique_keys(kp, n, ukeys.data_ptr(), first.data_ptr(), count.data_ptr(), ws.data_ptr(), need, _stream()))
    return ukeys, first, count


def coarsen(keys: torch.Tensor):
    """-> (parent_of[n], pkeys[n], child_row[n,8], count device int32[1]); only the first `count` parents are valid."""
    n = keys.shape[0]
    kp = _dev(keys, torch.int64, 'keys')
    dev = keys.device
    parent_of = torch.empty(n, dtype=torch.int32, device=dev)
    pkeys = torch.empty(n, dtype=torch.int64, device=dev)
    child_row = torch.empty((n, 8), dtype=torch.int32, device=dev)
    count = torch.empty(1, dtype=torch.int32, device=dev)
    L = lib()
    ws, need = _ws(lambda: L.fpcc_coarsen(None, n, None, None, None, None, None, 0, None), dev)
    _ok(L.fpcc_coarsen(kp, n, parent_of.data_ptr(), pkeys.data_ptr(), child_row.data_ptr(), count.data_ptr(),
                       ws.data_ptr(), need, _stream()))
    return parent_of, pkeys, child_row, count


def octree_level(keys: torch.Tensor, m: int, batch_shift: int) -> dict:
    """One level of the integer codec's encoder-side octree analysis (fpcc_octree_level): sorted duplicate-free keys [n] (Morton code
    with z on bit 0, sample index above it) and the row count m of the next coarser level -> dict(keys int64 [m], coords int32 [m, 4]
    = (sample, x, y, z), bits int32 [m, 8], table int32 [ceil128(m), 8] (child row + 1 | 0), symbols int16 [m])."""
    n, dev = keys.shape[0], keys.device
    rows = (m + 127) // 128 * 128
    out = {'keys': torch.empty(m, dtype=torch.int64, device=dev), 'coords': torch.empty((m, 4), dtype=torch.int32, device=dev),
           'bits': torch.empty((m, 8), dtype=torch.int32, device=dev), 'table': torch.empty((rows, 8), dtype=torch.int32, device=dev),
           'symbols': torch.empty(m, dtype=torch.int16, device=dev)}
    L = lib()
    ws, need = _ws(lambda: L.fpcc_octree_level(None, n, m, batch_shift, None, None, None, None, 0, None, None, 0, None), dev)
    _ok(L.fpcc_octree_level(_dev(keys, torch.int64, 'keys'), n, m, int(batch_shift), out['keys'].data_ptr(), out['coords'].data_ptr(),
                            out['bits'].data_ptr(), out['table'].data_ptr(), rows, out['symbols'].data_ptr(), ws.data_ptr(), need,
                            _stream()))
    return out


def level_counts(keys: torch.Tensor, levels: int) -> List[int]:
    """rows of the `levels` next coarser maps of a sorted key array: ONE kernel, ONE blocking 4 (levels + 1)-byte read-back
    (fpcc_level_histogram) instead of a read-back per fpcc_coarsen"""
    n = keys.shape[0]
    if n == 0:
        return [0] * levels
    hist = torch.empty(levels + 1, dtype=torch.int32, device=keys.device)
    _ok(lib().fpcc_level_histogram(_dev(keys, torch.int64, 'keys'), n, levels, hist.data_ptr(), _stream()))
    h = hist.tolist()
    out, acc = [], 0
    for t in range(levels, 0, -1):
        acc += h[t]
        out.append(1 + acc)
    return out[::-1]


def level_counts_clouds(keys: torch.Tensor, levels: int, cloud_shift: int, n_clouds: int) -> List[List[int]]:
    """rows of every cloud of a batch (key = cloud << cloud_shift | Morton code, rows cloud-major) on the map itself and on its
    `levels` next coarser maps: out[l][c], l = 0..levels.  One kernel, one blocking read-back (fpcc_level_histogram_clouds)."""
    n = keys.shape[0]
    hist = torch.empty((n_clouds, levels + 2), dtype=torch.int32, device=keys.device)
    _ok(lib().fpcc_level_histogram_clouds(_dev(keys, torch.int64, 'keys'), n, levels, int(cloud_shift), int(n_clouds),
                                          hist.data_ptr(), _stream()))
    h = hist.tolist()
    if sum(row[levels + 1] for row in h) != n:
        raise ValueError(f'a key carries a cloud index outside the batch of {n_clouds}: either the batch column is wrong or a coordinate '
                         f'(after subtracting its cloud\'s minimum) needs more than {cloud_shift // 3} bits per axis -- the Morton code of a '
                         f'cloud is {cloud_shift} bits wide (resolutions up to {1 << (cloud_shift // 3)})')
    out = [[0] * n_clouds for _ in range(levels + 1)]
    for c, row in enumerate(h):
        if row[levels + 1] and row[levels + 1] != 1 + sum(row[:levels + 1]):        # every neighbouring pair of distinct keys is in one bin
            raise ValueError(f'cloud {c}: the keys are not unique ({row[levels + 1]} keys, {1 + sum(row[:levels + 1])} distinct)')
        out[0][c] = row[levels + 1]
        acc = 0
        for t in range(levels, 0, -1):
            acc += row[t]
            out[t][c] = acc + (1 if row[levels + 1] else 0)
    return out


def refine(pkeys: torch.Tensor, mask: torch.Tensor):
    """mask uint8 [8m] -> (keys[8m], parent_of[8m], child_row[m,8], count device int32[1])"""
    m = pkeys.shape[0]
    dev = pkeys.device
    if mask.dtype == torch.bool:
        mask = mask.view(torch.uint8)
    if mask.numel() != 8 * m:
        raise ValueError('mask must have 8 entries per parent')
    keys = torch.empty(8 * m, dtype=torch.int64, device=dev)
    parent_of = torch.empty(8 * m, dtype=torch.int32, device=dev)
    child_row = torch.empty((m, 8), dtype=torch.int32, device=dev)
    count = torch.empty(1, dtype=torch.int32, device=dev)
    L = lib()
    ws, need = _ws(lambda: L.fpcc_refine(None, m, None, None, None, None, None, None, 0, None), dev)
    _ok(L.fpcc_refine(_dev(pkeys, torch.int64, 'pkeys'), m, _dev(mask, torch.uint8, 'mask'), keys.data_ptr(),
                      parent_of.data_ptr(), child_row.data_ptr(), count.data_ptr(), ws.data_ptr(), need, _stream()))
    return keys, parent_of, child_row, count


def nbr27_search(keys: torch.Tensor, bits: int) -> torch.Tensor:
    n = keys.shape[0]
    nbr = torch.empty((27, n), dtype=torch.int32, device=keys.device)
    _ok(lib().fpcc_nbr27_search(_dev(keys, torch.int64, 'keys'), n, bits, nbr.data_ptr(), _stream()))
    return nbr


def nbr27_from_parent(keys: Optional[torch.Tensor], parent_of: Optional[torch.Tensor], parent_nbr: torch.Tensor,
                      child_row: Optional[torch.Tensor], n: Optional[int] = None) -> torch.Tensor:
    """keys/parent_of/child_row all None: the full generated set of the parent level (n = 8 * parents)."""
    m = parent_nbr.shape[1]
    if keys is None:
        n = 8 * m if n is None else n
    else:
        n = keys.shape[0]
    nbr = torch.empty((27, n), dtype=torch.int32, device=parent_nbr.device)
    _ok(lib().fpcc_nbr27_from_parent(_dev(keys, torch.int64, 'keys', True), _dev(parent_of, torch.int32, 'parent_of', True),
                                     n, _dev(parent_nbr, torch.int32, 'parent_nbr'), m,
                                     _dev(child_row, torch.int32, 'child_row', True), nbr.data_ptr(), _stream()))
    return nbr


def nbr27_from_parent_ex(keys: Optional[torch.Tensor], parent_of: Optional[torch.Tensor], parent_nbr: torch.Tensor,
                         child_row: Optional[torch.Tensor], n: Optional[int] = None):
    """nbr27_from_parent plus, from the same pass, the table row-major [n, 32] and the rows' 27-bit presence masks:
    -> (nbr [27, n], rows [n, 32], masks int32 [n])"""
    m = parent_nbr.shape[1]
    n = (8 * m if n is None else n) if keys is None else keys.shape[0]
    dev = parent_nbr.device
    nbr = torch.empty((27, n), dtype=torch.int32, device=dev)
    rows = torch.empty((n, 32), dtype=torch.int32, device=dev)
    masks = torch.empty(n, dtype=torch.int32, device=dev)
    _ok(lib().fpcc_nbr27_from_parent_ex(_dev(keys, torch.int64, 'keys', True), _dev(parent_of, torch.int32, 'parent_of', True),
                                        n, _dev(parent_nbr, torch.int32, 'parent_nbr'), m,
                                        _dev(child_row, torch.int32, 'child_row', True), nbr.data_ptr(), rows.data_ptr(),
                                        masks.data_ptr(), _stream()))
    return nbr, rows, masks


def gather_table_rows(rows: torch.Tensor, order: torch.Tensor) -> torch.Tensor:
    """rows[order] of a row-major int32 table [n, ld] (ld % 4 == 0) as one launch of whole 16-byte pieces"""
    n, ld = rows.shape
    out = torch.empty_like(rows)
    _ok(lib().fpcc_gather_table_rows_i32(_dev(rows, torch.int32, 'rows'), ld, _dev(order, torch.int32, 'order'), n, out.data_ptr(), _stream()))
    return out


def mask27_from_parent(keys: Optional[torch.Tensor], parent_of: Optional[torch.Tensor], parent_nbr: torch.Tensor,
                       child_row: Optional[torch.Tensor], n: Optional[int] = None) -> torch.Tensor:
    """int32 [n]: bit d set where neighbour d of the row exists (nbr27_from_parent without the table)"""
    m = parent_nbr.shape[1]
    n = (8 * m if n is None else n) if keys is None else keys.shape[0]
    masks = torch.empty(n, dtype=torch.int32, device=parent_nbr.device)
    _ok(lib().fpcc_mask27_from_parent(_dev(keys, torch.int64, 'keys', True), _dev(parent_of, torch.int32, 'parent_of', True),
                                      n, _dev(parent_nbr, torch.int32, 'parent_nbr'), m,
                                      _dev(child_row, torch.int32, 'child_row', True), masks.data_ptr(), _stream()))
    return masks


def conv_ones_k3(masks: torch.Tensor, w: torch.Tensor, c_out: int, *, bias: Optional[torch.Tensor] = None, act: int = 0,
                 slope: Optional[torch.Tensor] = None, clip: float = 0.0) -> torch.Tensor:
    """3x3x3 convolution [27, 1, c_out] of the constant-one input from the rows' presence masks (fpcc_conv_ones_k3_f32)"""
    n = masks.shape[0]
    if w.numel() != 27 * c_out or not w.is_contiguous():
        raise ValueError('weights must be contiguous [27, 1, c_out]')
    out = torch.empty((n, c_out), dtype=torch.float32, device=masks.device)
    _ok(lib().fpcc_conv_ones_k3_f32(_dev(masks, torch.int32, 'masks'), n, _dev(w, torch.float32, 'w'),
                                    _dev(bias, torch.float32, 'bias', True), c_out, act, _dev(slope, torch.float32, 'slope', True),
                                    float(clip), out.data_ptr(), c_out, _stream()))
    return out


# ---------------------------------------------------------------------------------------------------------------
# convolution

def _rows2d(t: torch.Tensor, name: str):
    if t.dim() != 2 or t.dtype != torch.float32 or not t.is_cuda or t.stride(1) != 1:
        raise TypeError(f'{name} must be a 2-D float32 GPU tensor with unit column stride')
    return t.data_ptr(), t.shape[1], (t.stride(0) if t.shape[0] > 1 else max(t.shape[1], t.stride(0)))


# Packed copies of convolution weights for the wave-autonomous MFMA kernel (fpcc_conv_pack_weights_f32), one per weight
# tensor.  The entry keeps the source tensor alive (so its address cannot be handed to another tensor) and is rebuilt
# when the tensor was written in place (`_version`).  Inference only: training changes the weights every step.
_PACKED: 'collections.OrderedDict' = collections.OrderedDict()
_PACKED_MAX = 1024


def packed_weights(w: torch.Tensor, c1: int, c2: int, c_out: int, n_offsets: int, groups: int, fresh: bool = False) -> Optional[torch.Tensor]:
    """packed copy of w [groups, n_offsets, c1 + c2, c_out] for fpcc_conv_f32_pk, or None when the shape has no wave kernel.
    fresh: pack now and do not cache (weights that change every step: training)"""
    if not lib().fpcc_conv_packed_floats(c1, c2, c_out, n_offsets, groups):
        return None
    if fresh:
        out = torch.empty(w.numel(), dtype=torch.float32, device=w.device)
        _ok(lib().fpcc_conv_pack_weights_f32(w.data_ptr(), groups * n_offsets, c1 + c2, c_out, out.data_ptr(), _stream()))
        return out
    key = (w.data_ptr(), w.numel(), c1 + c2, c_out)
    ent = _PACKED.get(key)
    if ent is not None and ent[1] == w._version:
        _PACKED.move_to_end(key)
        return ent[2]
    out = torch.empty(w.numel(), dtype=torch.float32, device=w.device)
    _ok(lib().fpcc_conv_pack_weights_f32(w.data_ptr(), groups * n_offsets, c1 + c2, c_out, out.data_ptr(), _stream()))
    _PACKED[key] = (w, w._version, out)
    while len(_PACKED) > _PACKED_MAX:
        _PACKED.popitem(last=False)
    return out


KNOB_WAVE_ON, KNOB_WAVE_NBW, KNOB_WAVE_SB, KNOB_WAVE_DBG, KNOB_MFMA_TILE, KNOB_POINTWISE_ROWS = 0, 1, 2, 3, 5, 6
KNOB_GROUPED_FOLD_ROWS = 4       # rows from which grouped (order 3) layers run folded on one wave per unit; 0 = never
KNOB_GROUPED_OFF, KNOB_GROUPED_NBW, KNOB_WAVE22_ROWS = 7, 8, 9      # 7: experiments only (FPCC_EXPERIMENT=1), changes the summation order
KNOB_PERSIST = 12     # workgroups per CU of the persistent grouped / folded kernels (0 = off)
KNOB_LDS_ROWS, KNOB_LDS_ROW_BLOCKS = 10, 11   # rows from which order-3 layers take both operands through LDS (0 = never); row blocks per workgroup (2 | 3 | 4)


def numerics_version() -> int:
    """version of the summation-order rules that make up the stream format (include/fpcc_hip.h)"""
    return lib().fpcc_numerics_version()


def clock_probe(out2: torch.Tensor, spin_us: int = 20) -> None:
    """diagnostic (fpcc_clock_probe): enqueue a one-wave kernel that leaves (shader cycles, 100 MHz ticks) of a spin_us spin in the
    int64[2] device tensor out2"""
    _ok(lib().fpcc_clock_probe(_dev(out2, torch.int64, 'out2'), int(spin_us), _stream()))


def transpose_table(table: torch.Tensor, ld: int = 32) -> torch.Tensor:
    """offset-major neighbour table [n_offsets, n] -> row-major [n, ld] (pad -1): conv_f32(nbr=rows, nbr_ks=1, nbr_os=ld)"""
    k, n = table.shape
    out = torch.empty((n, ld), dtype=torch.int32, device=table.device)
    _ok(lib().fpcc_transpose_table_i32(_dev(table, torch.int32, 'table', n == 0), k, n, out.data_ptr(), ld, _stream()))
    return out


def conv_debug_stamps(buf: Optional[torch.Tensor]) -> None:
    """diagnostic (fpcc_conv_debug_stamps): int64 device buffer the grouped kernel's waves leave their stage stamps in while knob 3
    is 16; None detaches"""
    if buf is None:
        _ok(lib().fpcc_conv_debug_stamps(None, 0))
    else:
        _ok(lib().fpcc_conv_debug_stamps(_dev(buf, torch.int64, 'buf'), buf.numel()))


def conv_i8_debug_stamps(buf: Optional[torch.Tensor]) -> None:
    """attach (int64 device tensor) / detach (None) the stamp buffer of the int8 tiled convolution (fpcc_conv_i8_debug_stamps)"""
    if buf is None:
        _ok(lib().fpcc_conv_i8_debug_stamps(None, 0))
    else:
        _ok(lib().fpcc_conv_i8_debug_stamps(_dev(buf, torch.int64, 'buf'), buf.numel()))


def conv_set_tuning(which: int, value: int) -> int:
    """process-wide tuning knob of the wave kernel (fpcc_conv_set_tuning); returns the previous value"""
    before = _ok(lib().fpcc_conv_set_tuning(int(which), int(value)))
    conv_order.cache_clear()                    # knobs 4 / 7 (experiments only) move the summation-order thresholds
    return before


# When set to a list, every conv_f32 launch appends (start_event, end_event, info) recorded on the launch stream; used by
# bench.py to time the dominant kernel inside the timed region (events only, no synchronisation).
CONV_TRACE = None
TRACE_LOCK = __import__('threading').RLock()   # start event, launch, end event of a traced launch stay together when several threads enqueue
CLOCK_HOOK = None        # bench.py: callable(ev0, n_out, n_offsets) run right after a traced launch's start event (fpcc_clock_probe beside it)
_EVENT_POOL: list = []
_trace_tls = __import__('threading').local()


def set_thread_trace(trace) -> None:
    """a trace list for the launches of the CALLING thread only (None: back to the process-wide CONV_TRACE): with several frames in
    flight (fastpcc_amd/serving.py) a step is traced by tracing the thread that runs it"""
    global _n_thread_traces
    had = getattr(_trace_tls, 'trace', None) is not None
    _trace_tls.trace = trace
    with TRACE_LOCK:
        _n_thread_traces += (trace is not None) - had


_n_thread_traces = 0


def _untraced_launch(fn, call) -> None:
    """a launch of a thread that is not tracing: behind the lock while another thread is (its launch must not fall into a bracket)"""
    if _n_thread_traces:
        with TRACE_LOCK:
            _ok(fn(*call))
    else:
        _ok(fn(*call))


def _current_trace():
    t = getattr(_trace_tls, 'trace', None)
    return t if t is not None else CONV_TRACE


def reserve_trace_events(n: int) -> None:
    """create n timing events ahead of a traced step, so that the step itself records into existing events (creating an event
    per launch slows the host enough that the GPU waits for it, and those waits end up inside the measured intervals)"""
    while len(_EVENT_POOL) < n:
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()                               # creates the hipEvent_t behind it (see _trace_event)
        _EVENT_POOL.append(ev)


def _trace_event():
    if _EVENT_POOL:
        return _EVENT_POOL.pop()
    ev = torch.cuda.Event(enable_timing=True)
    ev.record()                                   # creates the hipEvent_t behind it (lazy in torch): fpcc_time_next_launch takes the handle
    return ev


def _traced_launch(trace: list, fn, call, info) -> None:
    """one launch with HIP events around it.  The events are recorded INSIDE the C call (fpcc_time_next_launch): recorded from here,
    whatever the interpreter does between the start event and the launch -- with several frames in flight, run another thread for a
    switch interval -- is timed as part of the kernel.  The lock keeps another thread's launch out of the bracket (libfpcc_hip's calls
    keep the interpreter lock -- _native.hip() -- but a thread may still be switched out between the two C calls below).  With a clock
    hook (bench.py's extra step, one thread) the start event is recorded here, the hook needs it.  If the launch raises, the pending
    bracket is withdrawn: it must not close around a later, unrelated launch of this thread."""
    with TRACE_LOCK:
        ev0, ev1 = _trace_event(), _trace_event()
        if CLOCK_HOOK is not None:
            ev0.record()
            CLOCK_HOOK(ev0, info.get('n_out', 0), info.get('n_offsets', 1))
            _ok(fn(*call))
            ev1.record()
        else:
            _ok(lib().fpcc_time_next_launch(ev0.cuda_event, ev1.cuda_event))
            try:
                _ok(fn(*call))
            except BaseException:
                lib().fpcc_time_next_launch(None, None)
                raise
        trace.append((ev0, ev1, info))


def conv_f32(x1: torch.Tensor, w: torch.Tensor, c_out: int, n_out: int, *, x2: Optional[torch.Tensor] = None,
             nbr: Optional[torch.Tensor] = None, n_offsets: int = 1, nbr_ks: int = 0, nbr_os: int = 1,
             bias: Optional[torch.Tensor] = None, groups: int = 1, out_map: Optional[torch.Tensor] = None,
             om_os: int = 0, om_gs: int = 1, out: Optional[torch.Tensor] = None, out_rows: Optional[int] = None,
             act: int = ACT_NONE, slope: Optional[torch.Tensor] = None, clip: float = 0.0,
             row_order: Optional[torch.Tensor] = None, pack: bool = False) -> torch.Tensor:
    """out[dst(o,g)] = act(sum_k X[nbr[k*nbr_ks + o*nbr_os]] @ w[g][k] + bias); see include/fpcc_hip.h.
    pack: True = keep a packed copy of `w` (packed_weights) and run the shapes that have one on the wave-autonomous / persistent
    per-point kernels -- for weights that stay put between calls (inference); 'fresh' = pack for this call only (training)."""
    p1, c1, ld1 = _rows2d(x1, 'x1')
    if x2 is not None:
        p2, c2, ld2 = _rows2d(x2, 'x2')
    else:
        p2, c2, ld2 = None, 0, 0
    if w.dtype != torch.float32 or not w.is_cuda or not w.is_contiguous() or \
            w.numel() != groups * n_offsets * (c1 + c2) * c_out:
        raise ValueError(f'weights must be contiguous fp32 [{groups},{n_offsets},{c1 + c2},{c_out}], got {tuple(w.shape)}')
    if out is None:
        rows = out_rows if out_rows is not None else n_out * groups
        out = torch.empty((rows, c_out), dtype=torch.float32, device=x1.device)
    po, co, ldo = _rows2d(out, 'out')
    if co != c_out:
        raise ValueError('output width mismatch')
    if om_os == 0:
        om_os = groups
    ws, ws_bytes = None, 0
    wp = packed_weights(w, c1, c2, c_out, n_offsets, groups, fresh=(pack == 'fresh')) if pack else None
    if wp is None and nbr is not None and n_offsets >= 8:      # a grouped (order 3) shape without a packed copy: room to pack into
        ws_bytes = lib().fpcc_conv_f32_ws_bytes(c1, c2, c_out, n_offsets, groups, n_out)
        if ws_bytes:
            ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device=x1.device)
    # arguments first, start event last: on a host-paced stretch the GPU reaches the event before the host has launched the kernel,
    # and whatever the host does in between (pointer checks, ctypes marshalling) would be timed as part of the launch
    fn = lib().fpcc_conv_f32_pk
    call = (p1, c1, ld1, p2, c2, ld2, _dev(nbr, torch.int32, 'nbr', True), n_offsets, nbr_ks, nbr_os,
            w.data_ptr(), None if wp is None else wp.data_ptr(),
            _dev(bias, torch.float32, 'bias', True), c_out, groups,
            _dev(out_map, torch.int32, 'out_map', True), om_os, om_gs, po, ldo, n_out, act,
            _dev(slope, torch.float32, 'slope', True), float(clip),
            _dev(row_order, torch.int32, 'row_order', True),
            None if ws is None else ws.data_ptr(), ws_bytes, _stream())
    trace = _current_trace()
    if trace is None:
        _untraced_launch(fn, call)
        return out
    _traced_launch(trace, fn, call, {'mfma': bool(conv_order(c1, c2, c_out, n_offsets, groups, n_out)), 'c_in': c1 + c2, 'c_out': c_out,
                                     'n_out': n_out, 'groups': groups, 'n_offsets': n_offsets, 'nbr': nbr,
                                     'nbr_ks': nbr_ks, 'nbr_os': nbr_os})
    return out


class _MlpLayer(C.Structure):
    _fields_ = [('w', _vp), ('w_packed', _vp), ('bias', _vp), ('slope', _vp), ('c_out', _i32), ('act', _i32), ('clip', _f32)]


class _MlpChain(C.Structure):
    _fields_ = [('x', _vp), ('cx', _i32), ('ldx', _i32), ('y', _vp), ('cy', _i32), ('ldy', _i32), ('cat_layer', _i32),
                ('n_layers', _i32), ('layers', _MlpLayer * 4), ('out', _vp), ('ldo', _i32), ('n', _i64)]


def pointwise_head_ok(c0: int, c1: int) -> bool:
    return (c0, c1) in ((16, 8), (8, 4))


def pointwise_head(x: torch.Tensor, w1: torch.Tensor, b1: Optional[torch.Tensor], act1: int, slope1: Optional[torch.Tensor], order1: int,
                   w2: torch.Tensor, b2: Optional[torch.Tensor], act2: int = ACT_NONE, slope2: Optional[torch.Tensor] = None,
                   clip: float = 0.0) -> torch.Tensor:
    """out [n, 1] = act2(act1(x @ w1 + b1) @ w2 + b2): the narrow two-layer head C0 -> C1 -> 1 as one launch (fpcc_pointwise_head_f32)"""
    px, c0, ldx = _rows2d(x, 'x')
    n = x.shape[0]
    if w1.dtype != torch.float32 or not w1.is_contiguous() or w1.shape[0] != c0 or w2.numel() != w1.shape[1] or not w2.is_contiguous():
        raise ValueError('weights must be contiguous fp32 [c0, c1] and [c1, 1]')
    out = torch.empty((n, 1), dtype=torch.float32, device=x.device)
    fn = lib().fpcc_pointwise_head_f32
    call = (px, c0, ldx, _dev(w1, torch.float32, 'w1'), _dev(b1, torch.float32, 'b1', True), w1.shape[1], int(act1),
            _dev(slope1, torch.float32, 'slope1', True), int(order1), _dev(w2, torch.float32, 'w2'),
            _dev(b2, torch.float32, 'b2', True), int(act2), _dev(slope2, torch.float32, 'slope2', True),
            float(clip), out.data_ptr(), n, _stream())
    trace = _current_trace()
    if trace is None:
        _untraced_launch(fn, call)
        return out
    _traced_launch(trace, fn, call, {'mfma': False, 'c_in': c0, 'c_out': 1, 'n_out': n, 'groups': 1, 'n_offsets': 1, 'nbr': None,
                                     'nbr_ks': 0, 'nbr_os': 1})
    return out


def mlp_chain_set_form(form: int) -> int:
    """1: workgroup form for the codecs' chain shapes (default), 0: wave form for all; returns the previous setting (result-neutral)"""
    return lib().fpcc_mlp_chain_set_form(int(form))


def mlp_chain_ok(cx: int, widths, cat_layer: int = -1, cy: int = 0) -> bool:
    """shapes fpcc_mlp_chain_f32 takes: 1..4 layers of width 32 | 64 | 128, input of 1 or a multiple of 32 (<= 256) channels, an
    optional concatenated operand of 32..128 channels entering a layer >= 1"""
    if not 1 <= len(widths) <= 4 or any(c not in (32, 64, 128) for c in widths):
        return False
    if not (cx == 1 or (cx % 32 == 0 and 32 <= cx <= 256)):
        return False
    if cat_layer == -1:
        return True
    return 1 <= cat_layer < len(widths) and cy % 32 == 0 and 32 <= cy <= 128


def mlp_chain(x: torch.Tensor, layers, y: Optional[torch.Tensor] = None, cat_layer: int = -1,
              out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """A stack of per-point layers as one launch (fpcc_mlp_chain_f32).  layers: sequence of (w [c_in, c_out] contiguous fp32,
    bias | None, act, slope | None, clip); `y` is concatenated after the activations entering layer `cat_layer`."""
    px, cx, ldx = _rows2d(x, 'x')
    n = x.shape[0]
    d = _MlpChain()
    d.x, d.cx, d.ldx, d.n, d.n_layers, d.cat_layer = px, cx, ldx, n, len(layers), cat_layer
    if y is not None:
        d.y, d.cy, d.ldy = _rows2d(y, 'y')
        if y.shape[0] != n:
            raise ValueError('concatenated operand has a different number of rows')
    keep = []
    c_in = cx
    flops = 0
    for l, (w, bias, act, slope, clip) in enumerate(layers):
        if l == cat_layer:
            c_in += d.cy
        if w.dtype != torch.float32 or not w.is_cuda or not w.is_contiguous() or w.dim() != 2 or w.shape[0] != c_in:
            raise ValueError(f'layer {l}: weights must be contiguous fp32 [{c_in}, c_out], got {tuple(w.shape)}')
        c_out = w.shape[1]
        L = d.layers[l]
        L.w = w.data_ptr()
        if c_in != 1:
            wp = packed_weights(w, c_in, 0, c_out, 1, 1)
            if wp is None:
                raise ValueError(f'layer {l}: {c_in} -> {c_out} has no packed form')
            keep.append(wp)
            L.w_packed = wp.data_ptr()
        L.bias = _dev(bias, torch.float32, 'bias', True)
        L.slope = _dev(slope, torch.float32, 'slope', True)
        L.c_out, L.act, L.clip = c_out, int(act), float(clip)
        flops += 2 * n * c_in * c_out
        c_in = c_out
    if out is None:
        out = torch.empty((n, c_in), dtype=torch.float32, device=x.device)
    d.out, co, d.ldo = _rows2d(out, 'out')
    if co != c_in or out.shape[0] != n:
        raise ValueError('output shape mismatch')
    fn, ref, stream = lib().fpcc_mlp_chain_f32, C.byref(d), _stream()
    trace = _current_trace()
    if trace is None:
        _untraced_launch(fn, (ref, stream))
        return out
    _traced_launch(trace, fn, (ref, stream), {'mfma': True, 'chain': True, 'c_in': cx, 'c_out': c_in, 'n_out': n, 'groups': 1, 'n_offsets': 1,
                                              'nbr': None, 'nbr_ks': 0, 'nbr_os': 1, 'flops': float(flops),
                                              'bytes': 4.0 * n * (cx + (d.cy if y is not None else 0) + c_in) + 4.0 * sum(w.numel() for w, *_ in layers),
                                              'layers': [tuple(w.shape) for w, *_ in layers]})
    return out


def nn_dist2(keys: torch.Tensor, bits: int, query: torch.Tensor, want_rows: bool = False):
    """squared distance from every query (batch, x, y, z) int32 row to its nearest voxel in the sorted key set (-1: none)"""
    n = query.shape[0]
    d = torch.empty(n, dtype=torch.int64, device=query.device)
    rows = torch.empty(n, dtype=torch.int32, device=query.device) if want_rows else None
    if query.dim() != 2 or query.shape[1] != 4:
        raise ValueError('query must be int32 [n, 4] = (batch, x, y, z)')
    _ok(lib().fpcc_nn_dist2(_dev(keys, torch.int64, 'keys', keys.numel() == 0), keys.shape[0], bits, _dev(query, torch.int32, 'query'), n,
                            d.data_ptr(), None if rows is None else rows.data_ptr(), _stream()))
    return (d, rows) if want_rows else d


def knn3d(p1: torch.Tensor, p2: torch.Tensor, K: int, version: int = -1):
    """lib.knn3d.knn3d: (idx int64 [P1, K], dist2 float32 [P1, K]) of the K nearest points of p2 for every point of p1"""
    if p1.dim() != 2 or p2.dim() != 2 or p1.shape[1] != 3 or p2.shape[1] != 3:
        raise ValueError('points must be [n, 3]')
    idx = torch.empty((p1.shape[0], K), dtype=torch.int64, device=p1.device)
    dist = torch.empty((p1.shape[0], K), dtype=torch.float32, device=p1.device)
    # the converted copies must outlive the launch: a temporary freed after taking its address can be handed out again by the
    # caching allocator for the second conversion
    p1f, p2f = p1.float().contiguous(), p2.float().contiguous()
    _ok(lib().fpcc_knn3d(_dev(p1f, torch.float32, 'p1', p1f.numel() == 0), p1f.shape[0],
                         _dev(p2f, torch.float32, 'p2', p2f.numel() == 0), p2f.shape[0], int(K),
                         idx.data_ptr(), dist.data_ptr(), _stream()))
    return idx, dist


def sum_i64(values: torch.Tensor) -> torch.Tensor:
    out = torch.empty(1, dtype=torch.int64, device=values.device)
    _ok(lib().fpcc_sum_i64(_dev(values, torch.int64, 'values', values.numel() == 0), values.numel(), out.data_ptr(), _stream()))
    return out


def knn_voxels(keys: torch.Tensor, bits: int, query: torch.Tensor, k: int, start_level: int = 1):
    """the k nearest voxels of every query row (batch, x, y, z) in the sorted key set, ordered by (squared distance, row):
    (rows int32 [n, k], dist2 int64 [n, k]); -1 where the set holds fewer than k voxels"""
    if query.dim() != 2 or query.shape[1] != 4:
        raise ValueError('query must be int32 [n, 4] = (batch, x, y, z)')
    n = query.shape[0]
    idx = torch.empty((n, k), dtype=torch.int32, device=query.device)
    d = torch.empty((n, k), dtype=torch.int64, device=query.device)
    _ok(lib().fpcc_knn_voxels(_dev(keys, torch.int64, 'keys', keys.numel() == 0), keys.shape[0], bits, _dev(query, torch.int32, 'query', n == 0),
                              n, int(k), int(start_level), idx.data_ptr(), d.data_ptr(), _stream()))
    return idx, d


def pca_normals(keys: torch.Tensor, bits: int, nbr: torch.Tensor) -> torch.Tensor:
    """unit normals float64 [n, 3] from the covariance of every point's neighbour rows `nbr` int32 [n, k] (fpcc_pca_normals)"""
    n, k = nbr.shape
    out = torch.empty((n, 3), dtype=torch.float64, device=nbr.device)
    _ok(lib().fpcc_pca_normals(_dev(keys, torch.int64, 'keys'), keys.shape[0], bits, _dev(nbr, torch.int32, 'nbr', n == 0), n, k,
                               out.data_ptr(), _stream()))
    return out


def nn_plane_dist2(keys: torch.Tensor, bits: int, normals: torch.Tensor, query: torch.Tensor):
    """(point-to-plane squared error float64 [n], nearest squared distance int64 [n], nearest row int32 [n]) of every query against
    the sorted key set whose rows carry `normals` float64 [m, 3]; ties at the nearest distance are averaged (fpcc_nn_plane_dist2)"""
    n = query.shape[0]
    plane = torch.empty(n, dtype=torch.float64, device=query.device)
    d = torch.empty(n, dtype=torch.int64, device=query.device)
    rows = torch.empty(n, dtype=torch.int32, device=query.device)
    _ok(lib().fpcc_nn_plane_dist2(_dev(keys, torch.int64, 'keys', keys.numel() == 0), keys.shape[0], bits,
                                  _dev(normals, torch.float64, 'normals', normals.numel() == 0), _dev(query, torch.int32, 'query', n == 0), n,
                                  d.data_ptr(), rows.data_ptr(), plane.data_ptr(), _stream()))
    return plane, d, rows


def transfer_normals(keys_a: torch.Tensor, coords_a: torch.Tensor, normals_a: torch.Tensor, keys_b: torch.Tensor, coords_b: torch.Tensor,
                     bits: int) -> torch.Tensor:
    """pc_error's normals for cloud B from those of cloud A (fpcc_transfer_normals); coords_*: (batch, x, y, z) of keys_* in row order"""
    n_a, n_b = keys_a.shape[0], keys_b.shape[0]
    out = torch.empty((n_b, 3), dtype=torch.float64, device=keys_b.device)
    L = lib()
    need = _ok(L.fpcc_transfer_normals_ws_bytes(n_a, n_b))
    ws = torch.empty(max(need // 8, 2), dtype=torch.int64, device=keys_b.device)
    _ok(L.fpcc_transfer_normals(_dev(keys_a, torch.int64, 'keys_a', n_a == 0), n_a, _dev(coords_a, torch.int32, 'coords_a', n_a == 0),
                                _dev(normals_a, torch.float64, 'normals_a', n_a == 0), _dev(keys_b, torch.int64, 'keys_b', n_b == 0), n_b,
                                _dev(coords_b, torch.int32, 'coords_b', n_b == 0), bits, out.data_ptr(), ws.data_ptr(), ws.numel() * 8, _stream()))
    return out


def sum_max_f64(values: torch.Tensor) -> torch.Tensor:
    """float64 [2] = (sum, maximum) of a float64 vector, evaluated in a fixed order (fpcc_sum_max_f64)"""
    n = values.numel()
    out = torch.empty(2, dtype=torch.float64, device=values.device)
    ws = torch.empty(max(2 * ((n + 4095) // 4096), 2), dtype=torch.float64, device=values.device)
    _ok(lib().fpcc_sum_max_f64(_dev(values, torch.float64, 'values', n == 0), n, out.data_ptr(), ws.data_ptr(), ws.numel() * 8, _stream()))
    return out


def conv_wgrad(x: torch.Tensor, dy: torch.Tensor, n: int, *, nbr: Optional[torch.Tensor] = None, n_offsets: int = 1,
               nbr_ks: int = 0, nbr_os: int = 1, out_map: Optional[torch.Tensor] = None, om_os: int = 0, om_gs: int = 1,
               groups: int = 1, out: Optional[torch.Tensor] = None, accumulate: bool = False,
               row_order: Optional[torch.Tensor] = None) -> torch.Tensor:
    """dW [groups, n_offsets, c_in, c_out] of the convolution whose forward used these row maps; see fpcc_conv_wgrad_f32"""
    px, c_in, ldx = _rows2d(x, 'x')
    pd, c_out, ldy = _rows2d(dy, 'dy')
    if out is None:
        out = torch.empty((groups, n_offsets, c_in, c_out), dtype=torch.float32, device=x.device)
        accumulate = False
    if not out.is_contiguous() or out.numel() != groups * n_offsets * c_in * c_out or out.dtype != torch.float32:
        raise ValueError('dw must be contiguous fp32 [groups, n_offsets, c_in, c_out]')
    L = lib()
    need = _ok(L.fpcc_conv_wgrad_ws_bytes(c_in, c_out, n_offsets, groups, n))
    ws = torch.empty(max(need // 4, 4), dtype=torch.float32, device=x.device)
    if om_os == 0:
        om_os = groups
    _ok(L.fpcc_conv_wgrad_f32(px, c_in, ldx, pd, c_out, ldy, _dev(nbr, torch.int32, 'nbr', True), n_offsets, nbr_ks, nbr_os,
                              _dev(out_map, torch.int32, 'out_map', True), om_os, om_gs, groups, n,
                              _dev(row_order, torch.int32, 'row_order', True), out.data_ptr(), int(accumulate), ws.data_ptr(), need,
                              _stream()))
    return out


def transpose_weights(w: torch.Tensor, n_offsets: int, c_in: int, c_out: int, flip: bool) -> torch.Tensor:
    """[n_offsets, c_in, c_out] -> [n_offsets, c_out, c_in] with the offsets mirrored when `flip` (input-gradient kernels)"""
    w = w.detach()
    if w.numel() != n_offsets * c_in * c_out or not w.is_contiguous():
        raise ValueError('weights must be contiguous [n_offsets, c_in, c_out]')
    wt = torch.empty((n_offsets, c_out, c_in), dtype=torch.float32, device=w.device)
    _ok(lib().fpcc_transpose_weights_f32(_dev(w, torch.float32, 'w'), n_offsets, c_in, c_out, int(flip), wt.data_ptr(), _stream()))
    return wt


def epilogue_bwd(y: torch.Tensor, dy: torch.Tensor, act: int, slope: Optional[torch.Tensor], want_bias: bool,
                 want_slope: bool):
    """-> (g, dbias [c] | None, dslope [1] | None): backward of conv_f32's fused bias + activation from the output y"""
    py, c, ldy = _rows2d(y, 'y')
    pd, c2, lddy = _rows2d(dy, 'dy')
    if c2 != c or y.shape[0] != dy.shape[0]:
        raise ValueError('y and dy must have one shape')
    n = y.shape[0]
    g = torch.empty((n, c), dtype=torch.float32, device=y.device)
    dbias = torch.empty(c, dtype=torch.float32, device=y.device) if want_bias else None
    dslope = torch.empty(1, dtype=torch.float32, device=y.device) if want_slope else None
    L = lib()
    need = _ok(L.fpcc_epilogue_bwd_ws_bytes(n, c))
    ws = torch.empty(max(need // 4, 4), dtype=torch.float32, device=y.device)
    _ok(L.fpcc_epilogue_bwd_f32(py, ldy, pd, lddy, n, c, act, _dev(slope, torch.float32, 'slope', True), g.data_ptr(), c,
                                None if dbias is None else dbias.data_ptr(), None if dslope is None else dslope.data_ptr(),
                                ws.data_ptr(), need, _stream()))
    return g, dbias, dslope


def deep_factorized_bits(y: torch.Tensor, weights, biases, factors, half_width: float, want_dy: bool = True):
    """y [n, c] fp32; parameters as stored by NoisyDeepFactorizedEntropyModel (filters 1-3-3-3-3-1).
    -> (out [c, 59]: parameter gradients of sum(logp) then sum(logp), dy [n, c] | None)"""
    py, c, ldy = _rows2d(y, 'y')
    n = y.shape[0]
    if len(weights) != 5 or len(biases) != 5 or len(factors) != 4:
        raise ValueError('the kernel is built for num_filters (1, 3, 3, 3, 3, 1)')
    shapes = [(c, 3, 1), (c, 3, 3), (c, 3, 3), (c, 3, 3), (c, 1, 3)]
    for t, sh in zip(weights, shapes):
        if tuple(t.shape) != sh:
            raise ValueError(f'weight of shape {tuple(t.shape)}, expected {sh}')
    ptr = lambda ts: (C.c_void_p * len(ts))(*[_dev(t.detach(), torch.float32, 'parameter') for t in ts])
    out = torch.empty((c, 59), dtype=torch.float32, device=y.device)
    dy = torch.empty((n, c), dtype=torch.float32, device=y.device) if want_dy else None
    L = lib()
    need = _ok(L.fpcc_deep_factorized_ws_bytes(n, c))
    ws = torch.empty(max(need // 4, 4), dtype=torch.float32, device=y.device)
    _ok(L.fpcc_deep_factorized_bits_f32(py, n, c, ldy, ptr(weights), ptr(biases), ptr(factors), float(half_width),
                                        None if dy is None else dy.data_ptr(), c, out.data_ptr(), ws.data_ptr(), need, _stream()))
    return out, dy


def noisy_normal_bits(y: torch.Tensor, index: torch.Tensor, log_scale_offset: float, log_scale_factor: float,
                      half_width: float = 0.5):
    """-> (sum of log-probabilities (0-dim), d/dy [n], d/dindex [n]) of y under N(0, exp(offset + factor * index)) + U(-h, h)"""
    n = y.numel()
    if index.numel() != n:
        raise ValueError('one index per value')
    index = index.to(torch.float32)
    py, pi = _dev(y, torch.float32, 'y', n == 0), _dev(index, torch.float32, 'index', n == 0)
    dy = torch.empty(n, dtype=torch.float32, device=y.device)
    di = torch.empty(n, dtype=torch.float32, device=y.device)
    out = torch.empty(1, dtype=torch.float32, device=y.device)
    L = lib()
    need = _ok(L.fpcc_noisy_normal_ws_bytes(n))
    ws = torch.empty(max(need // 4, 1), dtype=torch.float32, device=y.device)
    _ok(L.fpcc_noisy_normal_bits_f32(py, pi, n, float(log_scale_offset), float(log_scale_factor), float(half_width),
                                     dy.data_ptr(), di.data_ptr(), out.data_ptr(), ws.data_ptr(), need, _stream()))
    return out[0], dy, di


def conv_row_order(nbr: Optional[torch.Tensor], n_offsets: int, nbr_ks: int, nbr_os: int, n: int, window_log2: int = 13,
                   heaviest_first: bool = True, masks: Optional[torch.Tensor] = None) -> torch.Tensor:
    """permutation of the n output rows that puts rows with like neighbour patterns into the same MFMA block (windows of
    2^window_log2 rows) and, tile by tile, the tiles that execute most kernel offsets first; pass it to
    conv_f32(row_order=...) / conv_i8(row_order=...)"""
    L = lib()
    dev = masks.device if masks is not None else nbr.device
    keys = torch.empty(n, dtype=torch.int64, device=dev)
    group = 64 if n >= 32 * 1024 else 32              # the tile heights fpcc_conv_f32 uses for these map sizes
    n_groups = n // group
    # (on maps of millions of rows the launch has no tail worth shaping and the extra sort costs more than it returns)
    heaviest_first = heaviest_first and 2 <= n_groups <= LPT_MAX_GROUPS
    if masks is not None:                             # the rows' presence masks exist (the table's producer wrote them): 4 bytes per row
        _ok(L.fpcc_conv_row_keys_masks(_dev(masks, torch.int32, 'masks'), n, window_log2, keys.data_ptr(), _stream()))
    else:
        masks = torch.empty(n, dtype=torch.int32, device=dev) if heaviest_first else None
        _ok(L.fpcc_conv_row_keys(_dev(nbr, torch.int32, 'nbr'), n_offsets, nbr_ks, nbr_os, n, window_log2, keys.data_ptr(),
                                 None if masks is None else masks.data_ptr(), _stream()))
    order = sort_keys(keys, 32 + max(1, (n >> window_log2).bit_length()))[1]
    if not heaviest_first:
        return order
    gkeys = torch.empty(n_groups, dtype=torch.int64, device=dev)
    _ok(L.fpcc_conv_tile_keys(masks.data_ptr(), n_offsets, order.data_ptr(), n, group, gkeys.data_ptr(), _stream()))
    gperm = torch.empty(n_groups, dtype=torch.int32, device=dev)
    _ok(L.fpcc_conv_group_order(gkeys.data_ptr(), n_groups, gperm.data_ptr(), _stream()))
    out = torch.empty_like(order)
    _ok(L.fpcc_conv_regroup_rows(order.data_ptr(), gperm.data_ptr(), n, group, out.data_ptr(), _stream()))
    return out


LPT_MAX_GROUPS = 1 << 14

def table_conv_chunk(c_in: int, c_out: int, k: int) -> int:
    """Kernel offsets per launch of a general-table convolution (kernels other than 3x3x3 / 2x2x2, e.g. the 4x4x4 occupancy embedding
    with 64 offsets), whose launches' results are added in order.  fpcc_conv_f32 takes up to 32 offsets per launch on the VALU path
    and up to 27 on the MFMA path: shapes of the VALU path keep the partition into 32s they always had (their fp32 bits are pinned
    by fixtures), MFMA-capable shapes with more than 27 offsets go in 16s.  ONE rule for the inference partition
    (int_sparse_conv.Conv3d._run) and the training partition (autograd._forward), so that both sum in the same order."""
    mfma = conv_order(c_in, 0, c_out, 1, 1, 0) != 0
    if not mfma:
        return 32
    return k if k <= 27 else 16


@functools.lru_cache(maxsize=8192)
def conv_order(c1: int, c2: int, c_out: int, n_offsets: int = 1, groups: int = 1, n_out: int = 0) -> int:
    return lib().fpcc_conv_f32_order_ex(c1, c2, c_out, n_offsets, groups, n_out)


def gather_sum(y: torch.Tensor, nbr: torch.Tensor, n_offsets: int, nbr_ks: int, nbr_os: int, n: int, *,
               bias: Optional[torch.Tensor] = None, act: int = ACT_NONE, slope: Optional[torch.Tensor] = None,
               clip: float = 0.0) -> torch.Tensor:
    """out[o] = act(sum_k y[nbr[k*ks + o*os]][k] + bias): second half of a one-output-channel 3x3x3 convolution"""
    py, cy, ldy = _rows2d(y, 'y')
    out = torch.empty((n, 1), dtype=torch.float32, device=y.device)
    _ok(lib().fpcc_gather_sum_f32(py, ldy, _dev(nbr, torch.int32, 'nbr'), n_offsets, nbr_ks, nbr_os, n,
                                  _dev(bias, torch.float32, 'bias', True), act,
                                  _dev(slope, torch.float32, 'slope', True), float(clip), out.data_ptr(), _stream()))
    return out


def gather_sum_generated(y: torch.Tensor, parent_nbr: torch.Tensor, *, bias: Optional[torch.Tensor] = None, act: int = ACT_NONE,
                         slope: Optional[torch.Tensor] = None, clip: float = 0.0) -> torch.Tensor:
    """gather_sum on the generated set of a level (8 candidates per parent row) from the PARENT level's table [27, m]: the candidates'
    own 27-entry table is never built"""
    py, cy, ldy = _rows2d(y, 'y')
    m = parent_nbr.shape[1]
    if y.shape[0] != 8 * m:
        raise ValueError('y must have 8 rows per parent')
    out = torch.empty((8 * m, 1), dtype=torch.float32, device=y.device)
    _ok(lib().fpcc_gather_sum_generated_f32(py, ldy, _dev(parent_nbr, torch.int32, 'parent_nbr'), m, _dev(bias, torch.float32, 'bias', True),
                                            act, _dev(slope, torch.float32, 'slope', True), float(clip), out.data_ptr(), _stream()))
    return out


def gather_rows(x: torch.Tensor, index: torch.Tensor) -> torch.Tensor:
    """out[i] = x[index[i]] (rows; features re-ordered into canonical Morton row order)."""
    p, c, ld = _rows2d(x, 'x')
    n = index.shape[0]
    out = torch.empty((n, c), dtype=torch.float32, device=x.device)
    _ok(lib().fpcc_gather_rows_f32(p, c, ld, _dev(index, torch.int32, 'index'), n, out.data_ptr(), c, _stream()))
    return out


# ---------------------------------------------------------------------------------------------------------------
# entropy glue

def logit_to_prob16(logit: torch.Tensor) -> torch.Tensor:
    """uint16 probabilities, returned in an int16 tensor (same bits; view as uint16 on the host)."""
    n = logit.numel()
    out = torch.empty(n, dtype=torch.int16, device=logit.device)
    _ok(lib().fpcc_logit_to_prob16(_dev(logit, torch.float32, 'logit'), n, out.data_ptr(), _stream()))
    return out


def quantize_symbols_(x: torch.Tensor, scale: float = 1.0, want_symbols: bool = True) -> Optional[torch.Tensor]:
    n = x.numel()
    sym = torch.empty(n, dtype=torch.int32, device=x.device) if want_symbols else None
    _ok(lib().fpcc_quantize_symbols(_dev(x, torch.float32, 'x'), n, float(scale),
                                    None if sym is None else sym.data_ptr(), _stream()))
    return sym


def child_mask(child_row: torch.Tensor) -> torch.Tensor:
    m = child_row.shape[0]
    out = torch.empty(8 * m, dtype=torch.uint8, device=child_row.device)
    _ok(lib().fpcc_child_mask(_dev(child_row, torch.int32, 'child_row'), m, out.data_ptr(), _stream()))
    return out


def topk_keep(logit: torch.Tensor, target: int) -> torch.Tensor:
    n = logit.numel()
    if n % 8:
        raise ValueError('logits must come in groups of 8 children')
    m = n // 8
    out = torch.empty(n, dtype=torch.uint8, device=logit.device)
    L = lib()
    ws, need = _ws(lambda: L.fpcc_topk_keep(None, m, target, None, None, 0, None), logit.device)
    _ok(L.fpcc_topk_keep(_dev(logit, torch.float32, 'logit'), m, int(target), out.data_ptr(), ws.data_ptr(), need,
                         _stream()))
    return out


def topk_keep_cells(logit: torch.Tensor, cell_of_group: torch.Tensor, n_cells: int, target: int) -> torch.Tensor:
    """top-k keep where the local-maximum cell of candidate group p is cell_of_group[p] (see fpcc_topk_keep_cells)"""
    n = logit.numel()
    m = n // 8
    if n % 8 or cell_of_group.shape[0] != m:
        raise ValueError('one cell id per group of 8 candidates expected')
    out = torch.empty(n, dtype=torch.uint8, device=logit.device)
    L = lib()
    ws, need = _ws(lambda: L.fpcc_topk_keep_cells(None, m, None, n_cells, target, None, None, 0, None), logit.device)
    _ok(L.fpcc_topk_keep_cells(_dev(logit, torch.float32, 'logit'), m, _dev(cell_of_group, torch.int32, 'cell_of_group'),
                               n_cells, int(target), out.data_ptr(), ws.data_ptr(), need, _stream()))
    return out


def compact_coords(pkeys: torch.Tensor, mask: torch.Tensor, level: int, bits: int,
                   offset_xyz: Optional[torch.Tensor] = None):
    """Children (of parents pkeys at level+1) selected by mask -> xyz int32 [<=8m, 3] at `level`, + count (device)."""
    m = pkeys.shape[0]
    dev = pkeys.device
    if mask.dtype == torch.bool:
        mask = mask.view(torch.uint8)
    out = torch.empty((8 * m, 3), dtype=torch.int32, device=dev)
    count = torch.empty(1, dtype=torch.int32, device=dev)
    L = lib()
    ws, need = _ws(lambda: L.fpcc_compact_coords(None, m, None, level, bits, None, None, None, None, 0, None), dev)
    _ok(L.fpcc_compact_coords(_dev(pkeys, torch.int64, 'pkeys'), m, _dev(mask, torch.uint8, 'mask'), level, bits,
                              _dev(offset_xyz, torch.int32, 'offset', True), out.data_ptr(), count.data_ptr(),
                              ws.data_ptr(), need, _stream()))
    return out, count


# ---------------------------------------------------------------------------------------------------------------
# integer pipeline

def _any(t: Optional[torch.Tensor], name: str, dtypes, allow_none=False):
    if t is None:
        if allow_none:
            return None
        raise ValueError(f'{name} is required')
    if not t.is_cuda:
        raise FpccError(f'{name} must live on the GPU (libfpcc_hip has no CPU path)')
    if t.dtype not in dtypes:
        raise TypeError(f'{name} must be one of {dtypes}, got {t.dtype}')
    if not t.is_contiguous():
        raise ValueError(f'{name} must be contiguous')
    return t.data_ptr()


_U32 = (torch.uint32, torch.int32, torch.int64)


def _mul_u32(t: torch.Tensor) -> torch.Tensor:
    """requant multipliers live as uint32 in the library; checkpoints carry them as int64 (torch.save has no uint32)"""
    if t.dtype == torch.int64:
        return t.to(torch.int32) if int(t.max()) < 2 ** 31 else (t & 0xffffffff).to(torch.uint32)
    return t


def hash_insert_coords(table_keys: torch.Tensor, table_vals: torch.Tensor, coords_xyzb: torch.Tensor, batch_first: bool = False) -> None:
    """batch_first: rows are (batch, x, y, z) instead of the extension's (x, y, z, batch); same table either way"""
    fn = lib().fpcc_hash_insert_coords_bxyz if batch_first else lib().fpcc_hash_insert_coords
    _ok(fn(_dev(table_keys, torch.int64, 'table_keys'), _dev(table_vals, torch.int32, 'table_vals'),
                                      table_keys.shape[0], _dev(coords_xyzb, torch.int32, 'coords'), coords_xyzb.shape[0],
                                      _stream()))


def hash_lookup_coords(table_keys: torch.Tensor, table_vals: torch.Tensor, coords_xyzb: torch.Tensor, kernel_size,
                       stride, batch_first: bool = False, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """-> int32 [ceil(n/128)*128, volume], entry = input row + 1 or 0 (rows past n are zero), as the reference returns.
    out: a ZEROED int32 tensor of that shape to fill (a caller that clears it together with other buffers)"""
    n = coords_xyzb.shape[0]
    ks = (C.c_int32 * 3)(*[int(v) for v in kernel_size])
    st = (C.c_int32 * 3)(*[int(v) for v in stride])
    volume = int(kernel_size[0]) * int(kernel_size[1]) * int(kernel_size[2])
    rows = (n + 127) // 128 * 128
    if out is None:
        out = torch.zeros((rows, volume), dtype=torch.int32, device=coords_xyzb.device)
    elif out.dtype != torch.int32 or tuple(out.shape) != (rows, volume) or not out.is_contiguous() or not out.is_cuda:
        raise ValueError('out must be a contiguous int32 GPU tensor [ceil128(n), volume]')
    fn = lib().fpcc_hash_lookup_coords_bxyz if batch_first else lib().fpcc_hash_lookup_coords
    _ok(fn(_dev(table_keys, torch.int64, 'table_keys'), _dev(table_vals, torch.int32, 'table_vals'),
           table_keys.shape[0], _dev(coords_xyzb, torch.int32, 'coords'), n, ks, st, out.data_ptr(), _stream()))
    return out


def hash_insert_keys(table_keys, table_vals, keys) -> None:
    _ok(lib().fpcc_hash_insert_keys(_dev(table_keys, torch.int64, 'table_keys'), _dev(table_vals, torch.int32, 'table_vals'),
                                    table_keys.shape[0], _dev(keys, torch.int64, 'keys'), keys.shape[0], _stream()))


def hash_lookup_keys(table_keys, table_vals, keys) -> torch.Tensor:
    n = keys.shape[0]
    out = torch.zeros((n + 127) // 128 * 128, dtype=torch.int32, device=keys.device)
    _ok(lib().fpcc_hash_lookup_keys(_dev(table_keys, torch.int64, 'table_keys'), _dev(table_vals, torch.int32, 'table_vals'),
                                    table_keys.shape[0], _dev(keys, torch.int64, 'keys'), n, out.data_ptr(), _stream()))
    return out


def _rows_i8(t: torch.Tensor, name: str) -> torch.Tensor:
    """int8 [n, C] -> contiguous [n, ceil16(C)] with a zero pad when C is not a multiple of 16"""
    if t.dtype != torch.int8 or t.dim() != 2 or not t.is_cuda:
        raise TypeError(f'{name} must be a 2-D int8 GPU tensor')
    c = t.shape[1]
    if c % 16:
        return torch.nn.functional.pad(t, (0, 16 - c % 16)).contiguous()
    return t.contiguous()


class _Requant8(C.Structure):
    _fields_ = [('out', _vp), ('ld', _i32), ('pad', _i32), ('requant_mul', _vp), ('zero_point', _vp), ('shift', _i32)]


def _also8(also, n: int, c_out: int, device):
    """also: sequence of (requant_mul uint32[1], zero_point int64[1], shift, width) -- one extra int8 copy of an int32 result per
    consumer requantiser; width >= c_out is the row stride of the buffer (columns past c_out belong to the caller).  Returns the
    ctypes array, its length and the int8 tensors [n, width]."""
    if not also:
        return None, 0, []
    arr = (_Requant8 * len(also))()
    outs = []
    keep = []          # converted multipliers: a temporary freed before the launch could be handed out again by the caching allocator
    for i, (mul, zp, shift, width) in enumerate(also):
        buf = torch.empty((n, width), dtype=torch.int8, device=device)
        m = _mul_u32(mul)
        keep.append(m)
        arr[i].out, arr[i].ld, arr[i].pad = buf.data_ptr(), width, c_out
        arr[i].requant_mul, arr[i].zero_point, arr[i].shift = _any(m, 'requant_mul', _U32), _dev(zp, torch.int64, 'zero_point'), int(shift)
        outs.append(buf)
    arr._keep = keep   # lives as long as the descriptor array the caller passes to the launch
    return arr, len(also), outs


def fill_bits_i8(bits: torch.Tensor, out: torch.Tensor, col0: int, requant_mul: torch.Tensor, zero_point: torch.Tensor, shift: int,
                 fxp_one: int = 1 << 23) -> None:
    """out[:, col0:col0 + 8] = requant(bits ? fxp_one : 0); out[:, col0 + 8:] = 0   (fpcc_fill_bits_i8)"""
    n = bits.shape[0]
    if out.dtype != torch.int8 or out.dim() != 2 or out.shape[0] != n or out.stride(1) != 1:
        raise ValueError('out must be int8 [n, width]')
    _ok(lib().fpcc_fill_bits_i8(_dev(bits, torch.uint8, 'bits'), n, int(fxp_one), _any(_mul_u32(requant_mul), 'requant_mul', _U32),
                                _dev(zero_point, torch.int64, 'zero_point'), int(shift), out.data_ptr(), out.stride(0), int(col0),
                                out.shape[1], _stream()))


def conv_i8(a: torch.Tensor, w_padded: torch.Tensor, c_in: int, c_out: int, n_out: int, *,
            nbr: Optional[torch.Tensor] = None, n_offsets: int = 1, nbr_ks: int = 0, nbr_os: int = 1, nbr_bias: int = 0,
            zp_comp: Optional[torch.Tensor] = None, bias: Optional[torch.Tensor] = None,
            slope: Optional[torch.Tensor] = None, requant_mul: Optional[torch.Tensor] = None,
            zero_point: Optional[torch.Tensor] = None, shift: int = 0, out_bits: int = 32,
            row_order: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None,
            slope2: Optional[torch.Tensor] = None, also=None):
    """int8 sparse conv / linear with fused fixed-point epilogue; see fpcc_conv_i8.  w_padded: int8 [K, c_out, ldw].
    residual (int32 [n_out, c_out]) + slope2: out = prelu(residual + out) fused behind the epilogue (fpcc_conv_i8_res).
    also: extra int8 copies of an int32 result for its consumers' requantisers (see _also8): returns (out, [int8 tensors])."""
    a = _rows_i8(a, 'a')
    if not w_padded.is_cuda:
        raise FpccError('weights must live on the GPU (libfpcc_hip has no CPU path); move the model with .cuda()')
    if w_padded.dtype != torch.int8 or not w_padded.is_contiguous() or w_padded.dim() != 3 or \
            w_padded.shape[0] != n_offsets or w_padded.shape[1] != c_out or w_padded.shape[2] % 16:
        raise ValueError('weights must be contiguous int8 [n_offsets, c_out, ldw] with ldw a multiple of 16')
    out = torch.empty((n_out, c_out), dtype=torch.int8 if out_bits == 8 else torch.int32, device=a.device)
    mul = None if requant_mul is None else _mul_u32(requant_mul)
    ws, ws_bytes = None, 0
    if nbr is not None and n_offsets >= 8 and n_out <= 8192:
        ws_bytes = lib().fpcc_conv_i8_ws_bytes(1, n_offsets, int(mul is not None), c_out, n_out)
        if ws_bytes:
            ws = torch.empty(ws_bytes // 4, dtype=torch.int32, device=a.device)
    if residual is not None and tuple(residual.shape) != (n_out, c_out):
        raise ValueError('residual must be int32 [n_out, c_out]')
    arr, n_also, extra = _also8(also, n_out, c_out, a.device)
    _ok(lib().fpcc_conv_i8_also(a.data_ptr(), c_in, a.shape[1], _dev(nbr, torch.int32, 'nbr', True), n_offsets, nbr_ks, nbr_os,
                                nbr_bias, w_padded.data_ptr(), w_padded.shape[2], _dev(zp_comp, torch.int32, 'zp_comp', True),
                                _dev(bias, torch.int32, 'bias', True), _dev(slope, torch.int32, 'slope', True),
                                _any(mul, 'requant_mul', _U32, True), _dev(zero_point, torch.int64, 'zero_point', True),
                                int(shift), out_bits, out.data_ptr(), c_out, 0, c_out, n_out, _dev(row_order, torch.int32, 'row_order', True),
                                _dev(residual, torch.int32, 'residual', True), 0 if residual is None else c_out,
                                _dev(slope2, torch.int32, 'slope2', True), None if arr is None else C.cast(arr, _vp), n_also,
                                None if ws is None else ws.data_ptr(), ws_bytes, _stream()))
    return out if also is None else (out, extra)


def epilogue_i32(x: torch.Tensor, requant_mul: torch.Tensor, zero_point: Optional[torch.Tensor], shift: int, out_bits: int,
                 *, bias: Optional[torch.Tensor] = None, slope: Optional[torch.Tensor] = None,
                 row_group: Optional[torch.Tensor] = None, also=None):
    """requant_to_int8/int32 and their bias / PReLU variants on an int32 matrix [n, ch]; also: see conv_i8"""
    if x.dtype != torch.int32 or x.dim() != 2 or not x.is_cuda or x.stride(1) != 1:
        raise TypeError('input must be a 2-D int32 GPU tensor with unit column stride')
    n, ch = x.shape
    mul = _mul_u32(requant_mul)
    per_channel = 1 if mul.numel() == ch and ch > 1 else (1 if mul.numel() == ch else 0)
    if row_group is not None:
        if mul.numel() % ch or (bias is not None and bias.numel() != mul.numel()) or row_group.shape[0] != n:
            raise ValueError('grouped epilogue: requant_mul / bias hold groups x ch values, row_group one entry per row')
        per_channel = 1
    elif mul.numel() not in (1, ch):
        raise ValueError('requant_mul must have 1 or ch entries')
    out = torch.empty((n, ch), dtype=torch.int8 if out_bits == 8 else torch.int32, device=x.device)
    arr, n_also, extra = _also8(also, n, ch, x.device)
    _ok(lib().fpcc_epilogue_i32_also(x.data_ptr(), x.stride(0) if n > 1 else ch, _dev(bias, torch.int32, 'bias', True),
                                     _dev(slope, torch.int32, 'slope', True), _any(mul, 'requant_mul', _U32),
                                     per_channel, _dev(zero_point, torch.int64, 'zero_point', True), int(shift), out_bits,
                                     out.data_ptr(), ch, 0, n, ch, _dev(row_group, torch.int32, 'row_group', True),
                                     None if arr is None else C.cast(arr, _vp), n_also, _stream()))
    return out if also is None else (out, extra)


def octree_children(n: int, m: int, *, symbols: Optional[torch.Tensor] = None, bits: Optional[torch.Tensor] = None,
                    coords: Optional[torch.Tensor] = None, fxp_one: int = 1 << 23, want_table: bool = True, want_bits: bool = True):
    """One octree step of the integer codec (fpcc_octree_children): from the child occupancy of n parents (symbols int16 [n] or
    bits uint8 [n, 8]) with m occupied children in all -> dict(bits uint8 [n, 8], fxp int32 [n, 8], parent_row int32 [m],
    octant int32 [m], table int32 [ceil128(m), 8], child_coords int32 [m, 4] when the parents' coords are given)."""
    src = symbols if symbols is not None else bits
    dev = src.device
    out = {'parent_row': torch.empty(m, dtype=torch.int32, device=dev), 'octant': torch.empty(m, dtype=torch.int32, device=dev)}
    table_rows = (m + 127) // 128 * 128
    if want_table:
        out['table'] = torch.empty((table_rows, 8), dtype=torch.int32, device=dev)
    if want_bits:
        out['bits'] = torch.empty((n, 8), dtype=torch.uint8, device=dev)
        out['fxp'] = torch.empty((n, 8), dtype=torch.int32, device=dev)
    if coords is not None:
        if coords.shape != (n, 4):
            raise ValueError('coords must be int32 [n, 4]')
        out['child_coords'] = torch.empty((m, 4), dtype=torch.int32, device=dev)
    L = lib()
    ws, need = _ws(lambda: L.fpcc_octree_children(None, None, None, n, m, 0, None, None, None, None, 0, None, None, None, 0, None), dev)
    ptr = lambda k: out[k].data_ptr() if k in out else None
    _ok(L.fpcc_octree_children(_dev(symbols, torch.int16, 'symbols', True), _dev(bits, torch.uint8, 'bits', True),
                               _dev(coords, torch.int32, 'coords', True), n, m, int(fxp_one), ptr('child_coords'), ptr('parent_row'),
                               ptr('octant'), ptr('table'), table_rows if want_table else 0, ptr('bits'), ptr('fxp'),
                               ws.data_ptr(), need, _stream()))
    return out



# ---- integer codec: one level of the traversal per call (fpcc_int_level_trunk / fpcc_int_level_expand) ------------------------------------
class I8Layer(C.Structure):
    """fpcc_i8_layer"""
    _fields_ = [('w', _vp), ('ldw', _i32), ('c_in', _i32), ('c_out', _i32), ('n_offsets', _i32), ('zp_comp', _vp), ('bias', _vp),
                ('slope', _vp), ('requant_mul', _vp), ('zero_point', _vp), ('shift', _i32), ('out_bits', _i32)]


class I8Requant(C.Structure):
    """fpcc_i8_requant"""
    _fields_ = [('requant_mul', _vp), ('zero_point', _vp), ('shift', _i32)]


class IntOneScale(C.Structure):
    """fpcc_int_onescale"""
    _fields_ = [('channels', _i32), ('has_upsample', _i32),
                ('dec_in', I8Requant), ('dec_conv1', I8Layer), ('dec_conv2', I8Layer), ('dec_slope', _vp),
                ('pred_in', I8Requant), ('pred_conv', I8Layer), ('pred_linear', I8Layer),
                ('up_in', I8Requant), ('up_linear', I8Layer),
                ('up_res_in', I8Requant), ('up_conv1', I8Layer), ('up_conv2', I8Layer), ('up_slope', _vp),
                ('up_out_in', I8Requant), ('up_out', I8Layer)]


def i8_layer(w_padded: torch.Tensor, c_in: int, c_out: int, *, bias, slope, requant_mul, zero_point, shift: int, out_bits: int,
             zp_comp=None, keep: list) -> I8Layer:
    """descriptor of one int8 layer; the tensors it points at are appended to `keep` (the caller holds them as long as the descriptor)"""
    mul = _mul_u32(requant_mul)
    keep.extend(t for t in (w_padded, bias, slope, mul, zero_point, zp_comp) if t is not None)
    if w_padded.dtype != torch.int8 or not w_padded.is_contiguous() or w_padded.dim() != 3 or w_padded.shape[1] != c_out or w_padded.shape[2] % 16:
        raise ValueError('weights must be contiguous int8 [n_offsets, c_out, ldw] with ldw a multiple of 16')
    return I8Layer(w_padded.data_ptr(), w_padded.shape[2], c_in, c_out, w_padded.shape[0], _dev(zp_comp, torch.int32, 'zp_comp', True),
                   _dev(bias, torch.int32, 'bias', True), _dev(slope, torch.int32, 'slope', True), _any(mul, 'requant_mul', _U32),
                   _dev(zero_point, torch.int64, 'zero_point'), int(shift), int(out_bits))


def i8_requant(requant_mul: torch.Tensor, zero_point: torch.Tensor, shift: int, *, keep: list) -> I8Requant:
    mul = _mul_u32(requant_mul)
    keep.extend((mul, zero_point))
    return I8Requant(_any(mul, 'requant_mul', _U32), _dev(zero_point, torch.int64, 'zero_point'), int(shift))


_level_ws = {}          # (entry point, C, n, m) -> bytes: the size queries of the last levels seen (a frame asks for the same few again)


def int_level_trunk(blk: IntOneScale, n: int, feat: torch.Tensor, feat_q8: Optional[torch.Tensor], nbr27: torch.Tensor,
                    row_order: Optional[torch.Tensor], want_up: bool):
    """fpcc_int_level_trunk -> (res int32 [n, C], q_pred int8 [n, C], q_up int8 [n, ceil16(C + 8)] | None, logits int32 [n, 255])"""
    c, dev = blk.channels, feat.device
    if feat.dtype != torch.int32 or tuple(feat.shape) != (n, c) or not feat.is_contiguous():
        raise ValueError('feat must be contiguous int32 [n, channels]')
    if feat_q8 is not None and (feat_q8.dtype != torch.int8 or tuple(feat_q8.shape) != (n, c) or not feat_q8.is_contiguous()):
        raise ValueError('feat_q8 must be contiguous int8 [n, channels]')
    ld_up = (c + 8 + 15) // 16 * 16
    res = torch.empty((n, c), dtype=torch.int32, device=dev)
    q_pred = torch.empty((n, c), dtype=torch.int8, device=dev)
    q_up = torch.empty((n, ld_up), dtype=torch.int8, device=dev) if want_up else None
    logits = torch.empty((n, blk.pred_linear.c_out), dtype=torch.int32, device=dev)
    L, ref = lib(), C.byref(blk)
    key = ('trunk', c, n, feat_q8 is None)
    need = _level_ws.get(key)
    if need is None:
        if len(_level_ws) > 256:
            _level_ws.clear()
        need = _level_ws[key] = int(_ok(L.fpcc_int_level_trunk(ref, n, None, feat_q8.data_ptr() if feat_q8 is not None else None, None,
                                                              None, None, None, None, 0, None, None, 0, None)))
    ws = torch.empty(need, dtype=torch.uint8, device=dev)
    _ok(L.fpcc_int_level_trunk(ref, n, feat.data_ptr(), None if feat_q8 is None else feat_q8.data_ptr(), _dev(nbr27, torch.int32, 'nbr27'),
                               _dev(row_order, torch.int32, 'row_order', True), res.data_ptr(), q_pred.data_ptr(),
                               None if q_up is None else q_up.data_ptr(), ld_up, logits.data_ptr(), ws.data_ptr(), need, _stream()))
    return res, q_pred, q_up, logits


def int_level_expand(blk: IntOneScale, n: int, m: int, res: torch.Tensor, q_up: torch.Tensor, symbols: torch.Tensor,
                     coords: Optional[torch.Tensor], nbr27: torch.Tensor, row_order: Optional[torch.Tensor],
                     next_in: Optional[I8Requant]):
    """fpcc_int_level_expand -> (feat int32 [m, C], feat_q8 int8 [m, C] | None, child_coords int32 [m, 4] | None)"""
    c, dev = blk.channels, res.device
    if symbols.dtype != torch.int16 or symbols.shape[0] != n or not symbols.is_contiguous():
        raise ValueError('symbols must be contiguous int16 [n]')
    if q_up.dtype != torch.int8 or q_up.shape[0] != n or not q_up.is_contiguous() or tuple(res.shape) != (n, c) or not res.is_contiguous():
        raise ValueError('res / q_up must be the tensors int_level_trunk returned')
    feat = torch.empty((m, c), dtype=torch.int32, device=dev)
    feat_q8 = torch.empty((m, c), dtype=torch.int8, device=dev) if next_in is not None else None
    child = torch.empty((m, 4), dtype=torch.int32, device=dev) if coords is not None else None
    L, ref = lib(), C.byref(blk)
    key = ('expand', c, n, m)
    need = _level_ws.get(key)
    if need is None:
        if len(_level_ws) > 256:
            _level_ws.clear()
        need = _level_ws[key] = int(_ok(L.fpcc_int_level_expand(ref, n, m, None, None, 0, None, None, None, None, None, None, None, None,
                                                               None, 0, None)))
    ws = torch.empty(need, dtype=torch.uint8, device=dev)
    _ok(L.fpcc_int_level_expand(ref, n, m, res.data_ptr(), q_up.data_ptr(), q_up.shape[1], symbols.data_ptr(),
                                _dev(coords, torch.int32, 'coords', True), _dev(nbr27, torch.int32, 'nbr27'),
                                _dev(row_order, torch.int32, 'row_order', True), None if child is None else child.data_ptr(), feat.data_ptr(),
                                None if next_in is None else C.byref(next_in), None if feat_q8 is None else feat_q8.data_ptr(),
                                ws.data_ptr(), need, _stream()))
    return feat, feat_q8, child


def prelu_i32(a: torch.Tensor, slope: torch.Tensor, add: Optional[torch.Tensor] = None) -> torch.Tensor:
    a = a.contiguous()
    out = torch.empty_like(a)
    _ok(lib().fpcc_prelu_i32(_dev(a, torch.int32, 'a'), None if add is None else _dev(add.contiguous(), torch.int32, 'add'),
                             _dev(slope, torch.int32, 'slope'), a.numel(), out.data_ptr(), _stream()))
    return out


def softmax_i32(x: torch.Tensor) -> torch.Tensor:
    n, c = x.shape
    out = torch.empty((n, c), dtype=torch.uint32, device=x.device)
    _ok(lib().fpcc_softmax_i32(_dev(x, torch.int32, 'x'), n, c, out.data_ptr(), _stream()))
    return out


def logits_to_cdf16(logits: torch.Tensor, pre_shift: int) -> torch.Tensor:
    """uint16 CDF rows returned in an int16 tensor (same bits)"""
    n, c = logits.shape
    out = torch.empty((n, c), dtype=torch.int16, device=logits.device)
    _ok(lib().fpcc_logits_to_cdf16(_dev(logits, torch.int32, 'logits'), n, c, pre_shift, out.data_ptr(), _stream()))
    return out


def logits_to_ranges(logits: torch.Tensor, pre_shift: int, symbols: torch.Tensor):
    n, c = logits.shape
    start = torch.empty(n, dtype=torch.int16, device=logits.device)
    freqm1 = torch.empty(n, dtype=torch.int16, device=logits.device)
    _ok(lib().fpcc_logits_to_ranges(_dev(logits, torch.int32, 'logits'), n, c, pre_shift,
                                    _dev(symbols, torch.int16, 'symbols'), start.data_ptr(), freqm1.data_ptr(), _stream()))
    return start, freqm1


# ---------------------------------------------------------------------------------------------------------------
# device-side rANS decoders (one wave per stream)

def stream_to_device(data: bytes, device) -> torch.Tensor:
    """a byte stream as a uint8 device tensor (4 bytes of slack so that word-wise readers stay inside the allocation)"""
    t = torch.zeros(len(data) + 4, dtype=torch.uint8, pin_memory=True)
    t[:len(data)] = torch.frombuffer(bytearray(data), dtype=torch.uint8)
    return t.to(device, non_blocking=True)


def rans_binary_decode_dev(stream: torch.Tensor, stream_len: int, prob16: torch.Tensor):
    """-> (bits uint8 [n], ones int32 [1], status int32 [1]), all on the device; nothing is synchronised"""
    n = prob16.numel()
    dev = prob16.device
    bits = torch.empty(n, dtype=torch.uint8, device=dev)
    ones = torch.zeros(1, dtype=torch.int32, device=dev)
    status = torch.full((1,), -1, dtype=torch.int32, device=dev)
    _ok(lib().fpcc_rans_binary_decode_dev(_dev(stream, torch.uint8, 'stream'), int(stream_len), _dev(prob16, torch.int16, 'prob16', n == 0),
                                          n, bits.data_ptr(), ones.data_ptr(), status.data_ptr(), _stream()))
    return bits, ones, status


def simple_dec_pop_dev(state: torch.Tensor, stream: torch.Tensor, stream_len: int, rows: torch.Tensor):
    """rows int16 [n, width] (uint16 CDF rows of fpcc_logits_to_cdf16) -> (symbols int16 [n], int32 [2] = {children, sticky
    status word (0 = fine)}) on the device; `state` (int32 [4]) is advanced"""
    n, width = rows.shape
    sym = torch.empty(n, dtype=torch.int16, device=rows.device)
    children = torch.zeros(2, dtype=torch.int32, device=rows.device)
    _ok(lib().fpcc_simple_dec_pop_dev(_dev(state, torch.int32, 'state'), _dev(stream, torch.uint8, 'stream'), int(stream_len),
                                      _dev(rows, torch.int16, 'rows', n == 0), n, width, sym.data_ptr(), n, children.data_ptr(), _stream()))
    return sym, children
