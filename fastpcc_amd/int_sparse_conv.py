"""Integer-only sparse-conv operators: the Python surface of /root/reference/lib/int_sparse_conv/cuda_ops.py (same function
and class names, buffer names / dtypes / shapes -> same state_dict keys, same argument meaning) on top of libfpcc_hip.so.

    sparse_conv_in8w8out32, softmax_int32, GPUHashTable                       cuda_ops.py:95-169,684-685; binding.cu:114-145
    SparseConv{,PReLU}In8W8Out{8,32}, SparseResBlockIn32W8Out32               cuda_ops.py:62-92,189-455
    PReLUIn32Out32, RequantFxpToScaledInt8, Linear{,PReLU}In8W8Out{8,32}      cuda_ops.py:458-681
    SparseTensor (torchsparse's container: .F .C .stride .spatial_range ._caches)

Execution differs from the reference, results do not (exact integer arithmetic):
  * kernel maps stay the dense hash-lookup table [N, K]; the rulebook compaction with its host sync
    (cuda_ops.py:132-151) does not exist -- the convolution is output-stationary and reads the table directly;
  * convolution / linear, bias, PReLU and requantisation are ONE launch (fpcc_conv_i8) instead of up to 27 CUTLASS
    launches plus an epilogue kernel;
  * the float side of post-training quantisation lives here too: `Conv3d` (the float sparse convolution with
    torchsparse.nn.Conv3d's parameter names and layout, evaluated by fpcc_conv_f32 over the SAME lookup table as its
    integer counterpart), `make_obs` / `SparseTensorHistogramObserver` / `SparseResBlockWithObs` (cuda_ops.py:20-59);
    `import_parameters` applies the reference's conversion formulas (pinned by tests/golden/ptq_import.json).
"""
import functools
import math
from types import SimpleNamespace
from typing import Any, Dict, List, Optional, Tuple

import torch
import torch.nn as nn

from torch.ao.quantization import HistogramObserver

from . import hipops as ops

SharedFxpShift = 23               # Q8.23 shared fixed-point activation format
ROW_ORDER_MIN_ROWS = 8192         # larger 3x3x3 maps are evaluated in neighbour-pattern row order (smaller ones offset-split)
ROW_ORDER_WINDOW_LOG2 = 17
WeightRange = (1 << 7) - 1
ActRange = (1 << 7) - 1


class SparseTensor:
    """torchsparse.SparseTensor as the reference uses it: mutable feats / coords / stride plus a cache namespace shared by
    every tensor derived from the same cloud."""

    def __init__(self, feats: torch.Tensor, coords: torch.Tensor, stride=1, spatial_range=None):
        self.F = feats
        self.C = coords
        self.stride = (stride,) * 3 if isinstance(stride, int) else tuple(stride)
        self.spatial_range = spatial_range
        self._caches = SimpleNamespace(cmaps={}, kmaps={}, hashmaps={})

    @property
    def feats(self):
        return self.F

    @property
    def coords(self):
        return self.C


class GPUHashTable:
    """int_sparse_conv_ext.GPUHashTable(keys int64[cap], vals int32[cap]) over caller-owned tensors (binding.cu:134-145)"""

    def __init__(self, table_keys: torch.Tensor, table_vals: torch.Tensor):
        self.keys, self.vals = table_keys, table_vals

    def insert_coords(self, coords_xyzb: torch.Tensor) -> None:
        ops.hash_insert_coords(self.keys, self.vals, coords_xyzb)

    def lookup_coords(self, coords_xyzb: torch.Tensor, kernel_sizes, strides, kernel_volume: int) -> torch.Tensor:
        ks = kernel_sizes.tolist() if isinstance(kernel_sizes, torch.Tensor) else list(kernel_sizes)
        st = strides.tolist() if isinstance(strides, torch.Tensor) else list(strides)
        assert ks[0] * ks[1] * ks[2] == kernel_volume
        return ops.hash_lookup_coords(self.keys, self.vals, coords_xyzb, ks, st)

    def insert_vals(self, keys: torch.Tensor) -> None:
        ops.hash_insert_keys(self.keys, self.vals, keys)

    def lookup_vals(self, keys: torch.Tensor) -> torch.Tensor:
        return ops.hash_lookup_keys(self.keys, self.vals, keys)


def _pad_weight(weight: torch.Tensor) -> torch.Tensor:
    """int8 [K, C_out, C_in] -> contiguous [K, C_out, ceil16(C_in)] (zero padded): the row layout fpcc_conv_i8 reads"""
    c_in = weight.shape[-1]
    if c_in % 16:
        weight = torch.nn.functional.pad(weight, (0, 16 - c_in % 16))
    return weight.contiguous()


def _kernel_table(in_coords: torch.Tensor, out_coords: torch.Tensor, kernel_size, stride, hashmap_kv):
    """hash table of the input coordinates (built on first use) and the dense lookup table [N_out_padded, K] (row + 1 | 0)"""
    table = None
    if hashmap_kv is None:
        # keys, values and the lookup table are cleared by ONE fill (three small launches less per level of the integer codec)
        cap = 2 * in_coords.shape[0]
        rows, volume = (out_coords.shape[0] + 127) // 128 * 128, kernel_size[0] * kernel_size[1] * kernel_size[2]
        at = (12 * cap + 15) // 16 * 16
        buf = torch.zeros(at + 4 * rows * volume, dtype=torch.uint8, device=in_coords.device)
        keys, vals = buf[:8 * cap].view(torch.int64), buf[8 * cap:12 * cap].view(torch.int32)
        table = buf[at:].view(torch.int32).view(rows, volume)
        ops.hash_insert_coords(keys, vals, in_coords.contiguous(), batch_first=True)
        hashmap_kv = (keys, vals)
    table = ops.hash_lookup_coords(hashmap_kv[0], hashmap_kv[1], out_coords.contiguous(), kernel_size, stride, batch_first=True, out=table)
    return hashmap_kv, table


def sparse_conv_in8w8out32(in_feats: torch.Tensor, weight: torch.Tensor, in_coords: torch.Tensor, out_coords: torch.Tensor,
                           kernel_size: Tuple[int, int, int], stride: Tuple[int, int, int], in_out_maps=None,
                           hashmap_kv=None, zero_point_comp: Optional[torch.Tensor] = None,
                           if_in_coords_equals_out_coords: bool = False, _epilogue: Optional[dict] = None):
    """in_feats int8 [N1, C1], weight int8 [K, C2, C1], coords int32 [N, 4] = (batch, x, y, z) in level units.
    Returns (int32 [N2, C2] raw accumulators, hashmap_kv, in_out_maps) like the reference; `in_out_maps` is this
    build's kernel-map cache object (the dense lookup table), opaque to callers and accepted back as such."""
    volume = kernel_size[0] * kernel_size[1] * kernel_size[2]
    if in_out_maps is None:
        hashmap_kv, in_out_maps = _kernel_table(in_coords, out_coords, kernel_size, stride, hashmap_kv)
    n_out = out_coords.shape[0]
    if volume == 27 and n_out > ROW_ORDER_MIN_ROWS and not hasattr(in_out_maps, '_fpcc_row_order'):
        # neighbour-pattern row order, computed once per kernel map and kept with it (the table is the cache object)
        in_out_maps._fpcc_row_order = ops.conv_row_order(in_out_maps - 1, volume, 1, volume, n_out, ROW_ORDER_WINDOW_LOG2)
    w = weight if weight.shape[-1] % 16 == 0 and weight.is_contiguous() else _pad_weight(weight)
    ep = dict(_epilogue or {})
    hints = ep.pop('_hints', None)
    out = ops.conv_i8(in_feats, w, in_feats.shape[1], weight.shape[1], n_out, nbr=in_out_maps, n_offsets=volume, nbr_ks=1,
                      nbr_os=volume, nbr_bias=1, zp_comp=zero_point_comp, row_order=getattr(in_out_maps, '_fpcc_row_order', None), **ep)
    if ep.get('also') is not None:                  # (int32 result, its requantised int8 copies): keep them with the tensor
        out = _with_q8(out[0], hints, out[1])
    return out, hashmap_kv, in_out_maps


# ---- requantisation hints --------------------------------------------------------------------------------------------------
# A Q8.23 activation is requantised to int8 once per consumer (RequantFxpToScaledInt8, also the input_requant of a residual
# block), each a launch that reads the [n, C] int32 matrix again.  A producer that is told its consumers' requantisers (`_also`)
# writes those int8 copies from its own epilogue (fpcc_conv_i8_also / fpcc_epilogue_i32_also) and hangs them on the result
# tensor; the requantiser then finds its output ready.  Purely an execution detail: same integers, module tree untouched.
def _with_q8(t: torch.Tensor, hints, bufs) -> torch.Tensor:
    t._fpcc_q8 = {id(h): b for h, b in zip(hints, bufs)}
    return t


def _also_of(hints, c_out: int):
    """hints: RequantFxpToScaledInt8 modules, or (module, width) for a copy with a wider row (room for appended columns)"""
    mods, spec = [], []
    for h in hints or ():
        mod, width = h if isinstance(h, tuple) else (h, c_out)
        mods.append(mod)
        spec.append(mod.hint(width))
    return mods, spec


def _conv_on_sparse_tensor(input: 'SparseTensor', kernel_size, stride, run, unique=torch.unique) -> 'SparseTensor':
    """cache handling shared by the integer and float convolutions (cuda_ops.py:325-366): output coordinates, hash table and
    lookup table are taken from / stored in the cloud's caches; run(feats, in_coords, out_coords, maps, hashmap, same)
    -> (out_feats, hashmap, maps)"""
    caches = input._caches
    tag = (input.stride, kernel_size, stride)
    cur = caches.kmaps.get(tag)
    in_out_maps = cur.get('in_out_maps') if cur is not None else None
    hashmap_kv = caches.hashmaps.get(input.stride)
    if stride == (1, 1, 1):
        output_stride, output_coords, same = input.stride, input.C, True
    else:
        same = False
        output_stride = tuple(a * b for a, b in zip(input.stride, stride))
        if output_stride in caches.cmaps:
            output_coords = caches.cmaps[output_stride][0]
        elif stride[0] & (stride[0] - 1) == 0 and all(stride[0] == v for v in stride[1:]):
            output_coords = input.C.clone()
            output_coords[:, 1:] >>= (stride[0].bit_length() - 1)
            output_coords = unique(output_coords, dim=0)
        else:
            raise NotImplementedError((input.stride, stride))
    out_feats, hashmap_kv, in_out_maps = run(input.F, input.C, output_coords, in_out_maps, hashmap_kv, same)
    caches.kmaps.setdefault(tag, {}).setdefault('in_out_maps', in_out_maps)
    caches.hashmaps.setdefault(input.stride, hashmap_kv)
    caches.cmaps.setdefault(input.stride, (input.C, input.spatial_range))
    caches.cmaps.setdefault(output_stride, (output_coords, None))
    ret = SparseTensor(out_feats, output_coords, output_stride, None)
    ret._caches = caches
    return ret


# ---- float side of post-training quantisation (cuda_ops.py:20-59) ------------------------------------------------------
class Conv3d(nn.Module):
    """Float sparse convolution with torchsparse.nn.Conv3d's surface as the reference's float model uses it
    (models/convolutional/lossl_coord/model.py:31-35,648-650): parameters `kernel` [K, C_in, C_out] ([C_in, C_out] for a
    1x1x1 kernel) and `bias` [C_out]; attributes in_channels / out_channels / kernel_size / stride.  Offsets are ordered
    like the integer operator's weight[k], so `import_parameters` is a plain permute (cuda_ops.py:257-260)."""

    def __init__(self, in_channels: int, out_channels: int, kernel_size=3, stride=1, padding=0, bias: bool = True):
        super().__init__()
        as3 = lambda v: (v,) * 3 if isinstance(v, int) else tuple(v)
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride = as3(kernel_size), as3(stride)
        self.kernel_volume = math.prod(self.kernel_size)
        shape = (self.kernel_volume, in_channels, out_channels) if self.kernel_volume > 1 else (in_channels, out_channels)
        self.kernel = nn.Parameter(torch.zeros(shape))
        self.bias = nn.Parameter(torch.zeros(out_channels)) if bias else None
        std = 1.0 / math.sqrt(in_channels * self.kernel_volume)
        with torch.no_grad():
            self.kernel.uniform_(-std, std)
            if bias:
                self.bias.uniform_(-std, std)

    def extra_repr(self):
        return f'{self.in_channels}, {self.out_channels}, kernel_size={self.kernel_size}, stride={self.stride}'

    def _run(self, in_feats, in_coords, out_coords, in_out_maps, hashmap_kv, same):
        if in_out_maps is None:
            hashmap_kv, in_out_maps = _kernel_table(in_coords, out_coords, self.kernel_size, self.stride, hashmap_kv)
        nbr = in_out_maps - 1                      # (row + 1 | 0) -> (row | -1)
        k, n_out = self.kernel_volume, out_coords.shape[0]
        w = self.kernel.detach().reshape(k, self.in_channels, self.out_channels)
        x = in_feats.float().contiguous()
        bias = None if self.bias is None else self.bias.detach()
        out = None
        step = ops.table_conv_chunk(self.in_channels, self.out_channels, k)
        for a in range(0, k, step):                # fpcc_conv_f32 takes up to 32 (MFMA path: 27) offsets per launch (a 4x4x4 kernel has 64)
            b = min(a + step, k)
            part = ops.conv_f32(x, w[a:b].reshape(1, b - a, self.in_channels, self.out_channels).contiguous(), self.out_channels,
                                n_out, nbr=nbr if (a, b) == (0, k) else nbr[:, a:b].contiguous(), n_offsets=b - a, nbr_ks=1,
                                nbr_os=b - a, bias=bias if a == 0 else None)
            out = part if out is None else out.add_(part)
        return out, hashmap_kv, in_out_maps

    def _run_autograd(self, in_feats, in_coords, out_coords, in_out_maps, hashmap_kv, same):
        """training path: the same lookup table, the convolution as an autograd function of fastpcc_amd/autograd.py (weight
        and input gradients on the MFMA kernels); 3x3x3 on one coordinate set and 2x2x2 / stride 2 are what the models use"""
        from .autograd import ConvSpec, sparse_conv
        if in_out_maps is None:
            with torch.no_grad():
                hashmap_kv, in_out_maps = _kernel_table(in_coords, out_coords, self.kernel_size, self.stride, hashmap_kv)
        n_in, n_out = in_coords.shape[0], out_coords.shape[0]
        if self.kernel_size == (3, 3, 3) and same:
            if not hasattr(in_out_maps, '_fpcc_offset_major'):       # [27][n] (row | -1): the layout the gradient kernels walk
                in_out_maps._fpcc_offset_major = (in_out_maps[:n_out] - 1).t().contiguous()
            spec = ConvSpec('k3', n_in, n_out, in_out_maps._fpcc_offset_major)
        elif self.kernel_size == (2, 2, 2) and self.stride == (2, 2, 2):
            if not hasattr(in_out_maps, '_fpcc_child_row'):
                in_out_maps._fpcc_child_row = (in_out_maps[:n_out] - 1).contiguous()
            spec = ConvSpec('k2s2', n_in, n_out, in_out_maps._fpcc_child_row)
        else:
            # any other kernel (the 4x4x4 / stride-4 embedding of occupancy bits): weight gradient only
            if in_feats.requires_grad:
                raise NotImplementedError(f'input gradient through a {self.kernel_size} / stride {self.stride} convolution')
            if not hasattr(in_out_maps, '_fpcc_rows'):
                in_out_maps._fpcc_rows = (in_out_maps[:n_out] - 1).contiguous()
            spec = ConvSpec('tab', n_in, n_out, in_out_maps._fpcc_rows)
        w = self.kernel.reshape(self.kernel_volume, self.in_channels, self.out_channels)
        out = sparse_conv(in_feats.float(), w, spec)
        if self.bias is not None:
            out = out + self.bias
        return out, hashmap_kv, in_out_maps

    def forward(self, input: 'SparseTensor') -> 'SparseTensor':
        if torch.is_grad_enabled() and (self.kernel.requires_grad or input.F.requires_grad):
            return _conv_on_sparse_tensor(input, self.kernel_size, self.stride, self._run_autograd, unique=torch.unique_consecutive)
        with torch.no_grad():
            return _conv_on_sparse_tensor(input, self.kernel_size, self.stride, self._run, unique=torch.unique_consecutive)


class SparseTensorHistogramObserver(HistogramObserver):
    """torch.ao's histogram observer fed with a sparse tensor's features (cuda_ops.py:20-33)"""

    def forward(self, input):
        super().forward(input.F if isinstance(input, SparseTensor) else input)
        return input

    def extra_repr(self):
        return f'min_val={self.min_val}, max_val={self.max_val}, {self.qscheme}'


def make_obs(qscheme=torch.per_tensor_symmetric):
    return SparseTensorHistogramObserver(bins=2048, dtype=torch.qint8, quant_min=-ActRange, quant_max=ActRange, qscheme=qscheme)


class SparseResBlockWithObs(nn.Module):
    """float residual block with observers in front of both convolutions (cuda_ops.py:41-59)"""

    def __init__(self, ch: int):
        super().__init__()
        self.ch = ch
        self.obs = make_obs(torch.per_tensor_symmetric)
        self.conv = Conv3d(ch, ch, 3, 1, 1, bias=True)
        self.act = nn.PReLU()
        self.obs2 = make_obs(torch.per_tensor_symmetric)
        self.conv2 = Conv3d(ch, ch, 3, 1, 1, bias=True)
        self.act2 = nn.PReLU()

    def forward(self, org: 'SparseTensor') -> 'SparseTensor':
        org = self.obs(org)
        x = self.conv(org)
        x.F = self.act(x.F)
        x = self.obs2(x)
        x = self.conv2(x)
        x.F = self.act2(x.F + org.F)
        return x


def softmax_int32(input: torch.Tensor) -> torch.Tensor:
    return ops.softmax_i32(input.contiguous())


class LoadSaveUint32RequantMul(nn.Module):
    """torch.save has no uint32: `requant_mul` is stored as int64 and loaded back as uint32 (cuda_ops.py:172-186)."""

    def _save_to_state_dict(self, destination, prefix, keep_vars):
        super()._save_to_state_dict(destination, prefix, keep_vars)
        key = prefix + 'requant_mul'
        destination[key] = destination[key].to(torch.int64)

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        key = prefix + 'requant_mul'
        if key in state_dict:
            state_dict[key] = state_dict[key].to(torch.uint32)
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs)


def _shift_and_mul(requant_mul: torch.Tensor, guard_bits: int):
    shift = torch.log2((1 << (32 - guard_bits)) / requant_mul).min().floor()
    if shift < 0:
        raise ValueError(f'negative requant shift {shift}')
    return shift.to(torch.int32), (requant_mul * 2 ** shift.to(torch.float)).round().to(torch.int64)


class _RequantParams(LoadSaveUint32RequantMul):
    """buffers shared by the conv and linear operators (cuda_ops.py:194-206,516-528)"""

    def _make_buffers(self, weight_shape, out_ch, with_prelu):
        self.register_buffer('weight', torch.zeros(weight_shape, dtype=torch.int8), persistent=True)
        self.register_buffer('bias', torch.zeros((out_ch,), dtype=torch.int32), persistent=True)
        if with_prelu:
            self.register_buffer('slope', torch.zeros((1,), dtype=torch.int32), persistent=True)
        self.register_buffer('requant_mul', torch.zeros((out_ch,), dtype=torch.uint32), persistent=True)
        self.register_buffer('requant_shift', torch.zeros((1,), dtype=torch.int32), persistent=True)
        self.register_buffer('int_zero_point_out', torch.zeros((1,), dtype=torch.int64), persistent=True)
        self.register_buffer('scale_in', torch.zeros((1,), dtype=torch.float32) - 1, persistent=True)
        self.register_buffer('zero_point_in', torch.zeros((1,), dtype=torch.float32), persistent=True)
        self.register_buffer('scale_weight', torch.zeros((out_ch,), dtype=torch.float32) - 1, persistent=True)
        self.register_buffer('scale_out', torch.zeros((1,), dtype=torch.float32) - 1, persistent=True)
        self.register_buffer('zero_point_out', torch.zeros((1,), dtype=torch.float32), persistent=True)
        self._shift_host: Optional[int] = None
        self._w_pad = None
        self._w_tag = None

    def _shift(self) -> int:
        if self._shift_host is None:
            self._shift_host = int(self.requant_shift.item())
        return self._shift_host

    def _padded_weight(self) -> torch.Tensor:
        tag = (self.weight._version, self.weight.data_ptr())
        if self._w_tag != tag:
            w = self.weight if self.weight.dim() == 3 else self.weight.unsqueeze(0)
            self._w_pad, self._w_tag = _pad_weight(w), tag
        return self._w_pad

    def _load_from_state_dict(self, *args, **kwargs):
        self._shift_host = None
        super()._load_from_state_dict(*args, **kwargs)

    def _epilogue(self) -> dict:
        shift = self._shift() if self.out_scaled_int else self._shift() - SharedFxpShift
        return dict(bias=self.bias, slope=self.slope if self.with_prelu else None, requant_mul=self.requant_mul,
                    zero_point=self.int_zero_point_out, shift=shift, out_bits=8 if self.out_scaled_int else 32)

    @torch.no_grad()
    def _import_common(self, scale_in, zero_point_in, scale_out, zero_point_out, w_float_out_major, bias_float,
                       prelu_weight, fold_zero_point_into_bias: bool):
        """the conversion of cuda_ops.py:223-301 / 542-607; w_float_out_major: [.., C_out, C_in] with C_out at dim -2"""
        eps, dev = self.eps, self.weight.device          # observers may have been moved to the host for the parameter search
        scale_in, zero_point_in = scale_in.to(dev), zero_point_in.to(dev)
        w_float_out_major, bias_float = w_float_out_major.to(dev), bias_float.to(dev)
        scale_in = scale_in.reshape(1).float().clip(min=eps)
        self.scale_in[:] = scale_in
        if self.out_scaled_int:
            scale_out, zero_point_out = scale_out.to(dev), zero_point_out.to(dev)
            scale_out = scale_out.reshape(1).float().clip(min=eps)
            self.scale_out[:] = scale_out
        elif scale_out is not None:
            raise ValueError('fixed-point outputs take no output scale')
        reduce_dims = tuple(d for d in range(w_float_out_major.dim()) if d != w_float_out_major.dim() - 2)
        scale_weight = (w_float_out_major.abs().amax(dim=reduce_dims) / WeightRange).clip(min=eps)
        self.scale_weight[:] = scale_weight
        shape = [1] * w_float_out_major.dim()
        shape[-2] = -1
        self.weight[...] = (w_float_out_major / scale_weight.reshape(shape)).round().clip(-WeightRange, WeightRange).to(torch.int8)
        zp_in = zero_point_in.reshape(1)
        self.zero_point_in[:] = zp_in
        scale_bias = scale_in * scale_weight
        bias = bias_float / scale_bias
        if fold_zero_point_into_bias:
            bias = bias - (zp_in.to(torch.float) * self.weight.to(torch.float)).sum(-1).reshape(-1)
        self.bias[:] = bias.round().to(torch.int32)
        if self.with_prelu:
            if prelu_weight.numel() != 1:
                raise ValueError('single-parameter PReLU expected')
            self.slope[:] = (prelu_weight.to(dev).reshape(1) * (1 << 25)).round().to(torch.int32)
        requant = scale_in * scale_weight / scale_out if self.out_scaled_int else scale_in * scale_weight
        shift, mul = _shift_and_mul(requant, self.requant_mul_guard_bits)
        self.requant_shift[:] = shift
        self.requant_mul[:] = mul.to(torch.uint32)
        if self.out_scaled_int:
            self.zero_point_out[:] = zero_point_out.reshape(1)
            self.int_zero_point_out[:] = zero_point_out.reshape(1).to(torch.int64) << int(shift)
        else:
            self.zero_point_out[:] = 0
            self.int_zero_point_out[:] = 0
        self._shift_host = None
        return zp_in


class SparseConvIn8Out8(_RequantParams):
    def __init__(self, in_ch: int, out_ch: int, kernel_size: Tuple[int, int, int], stride: Tuple[int, int, int],
                 with_prelu: bool, out_scaled_int: bool, eps=None, requant_mul_guard_bits=None):
        super().__init__()
        self.kernel_volume = kernel_size[0] * kernel_size[1] * kernel_size[2]
        self._make_buffers((self.kernel_volume, out_ch, in_ch), out_ch, with_prelu)
        self.in_ch, self.out_ch = in_ch, out_ch
        self.kernel_size, self.stride = tuple(kernel_size), tuple(stride)
        self.with_prelu, self.out_scaled_int = with_prelu, out_scaled_int
        self.use_zero_point_in = False
        self.eps = eps if eps is not None else torch.finfo(torch.float32).eps
        self.requant_mul_guard_bits = requant_mul_guard_bits if requant_mul_guard_bits is not None else 10

    @torch.no_grad()
    def import_parameters(self, scale_in, zero_point_in, scale_out, zero_point_out, conv, prelu):
        """conv: object with .kernel [K, C_in, C_out] float and .bias; prelu: object with .weight or None"""
        kernel = conv.kernel.detach().float().reshape(self.kernel_volume, self.in_ch, self.out_ch)
        zp_in = self._import_common(scale_in, zero_point_in, scale_out, zero_point_out, kernel.permute(0, 2, 1),
                                    conv.bias.detach().float().reshape(-1), None if prelu is None else prelu.weight.detach().float(),
                                    fold_zero_point_into_bias=False)
        if int(zp_in.item()) != 0:
            self.use_zero_point_in = True
            self.register_buffer('int_zero_point_in_comp', torch.zeros((self.kernel_volume, self.out_ch), dtype=torch.int32,
                                                                       device=self.weight.device), persistent=True)
            self.int_zero_point_in_comp[...] = -(zp_in.to(torch.float) * self.weight.to(torch.float)).sum(2).round().to(torch.int32)

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        if prefix + 'int_zero_point_in_comp' in state_dict:
            self.use_zero_point_in = True
            self.register_buffer('int_zero_point_in_comp', torch.zeros((self.kernel_volume, self.out_ch), dtype=torch.int32),
                                 persistent=True)
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs)

    def unique(self, *args, **kwargs):
        return torch.unique(*args, **kwargs)

    def forward(self, *args, **kwargs):
        first = args[0] if args else kwargs.get('input')
        if isinstance(first, SparseTensor):
            return self.forward_with_sparse_tensor(*args, **kwargs)
        return self.forward_with_coords(*args, **kwargs)

    def forward_with_sparse_tensor(self, input: SparseTensor, _residual=None, _also=None) -> SparseTensor:
        run = self.forward_with_coords if _residual is None and not _also else \
            functools.partial(self.forward_with_coords, _residual=_residual, _also=_also)
        return _conv_on_sparse_tensor(input, self.kernel_size, self.stride, run, self.unique)

    def forward_with_coords(self, in_feats, in_coords, out_coords, in_out_maps=None, hashmap_kv=None,
                            if_in_coords_equals_out_coords: bool = False, _residual=None, _also=None):
        ep = self._epilogue()
        if _residual is not None:
            ep['residual'], ep['slope2'] = _residual
        if _also and not self.out_scaled_int:
            ep['_hints'], ep['also'] = _also_of(_also, self.out_ch)
        return sparse_conv_in8w8out32(
            in_feats, self._padded_weight(), in_coords, out_coords, self.kernel_size, self.stride, in_out_maps, hashmap_kv,
            self.int_zero_point_in_comp if self.use_zero_point_in else None, if_in_coords_equals_out_coords,
            _epilogue=ep)


class SparseConvIn8W8Out8(SparseConvIn8Out8):
    def __init__(self, in_ch, out_ch, kernel_size=(3, 3, 3), stride=(1, 1, 1), *args, **kwargs):
        super().__init__(in_ch, out_ch, kernel_size, stride, False, True, *args, **kwargs)

    def import_parameters(self, scale_in, zero_point_in, scale_out, zero_point_out, conv):
        super().import_parameters(scale_in, zero_point_in, scale_out, zero_point_out, conv, None)


class SparseConvIn8W8Out32(SparseConvIn8Out8):
    def __init__(self, in_ch, out_ch, kernel_size=(3, 3, 3), stride=(1, 1, 1), *args, **kwargs):
        super().__init__(in_ch, out_ch, kernel_size, stride, False, False, *args, **kwargs)

    def import_parameters(self, scale_in, zero_point_in, conv):
        super().import_parameters(scale_in, zero_point_in, None, None, conv, None)


class SparseConvPReLUIn8W8Out8(SparseConvIn8Out8):
    def __init__(self, in_ch, out_ch, kernel_size=(3, 3, 3), stride=(1, 1, 1), *args, **kwargs):
        super().__init__(in_ch, out_ch, kernel_size, stride, True, True, *args, **kwargs)

    def import_parameters(self, scale_in, zero_point_in, scale_out, zero_point_out, conv, prelu):
        super().import_parameters(scale_in, zero_point_in, scale_out, zero_point_out, conv, prelu)


class SparseConvPReLUIn8W8Out32(SparseConvIn8Out8):
    def __init__(self, in_ch, out_ch, kernel_size=(3, 3, 3), stride=(1, 1, 1), *args, **kwargs):
        super().__init__(in_ch, out_ch, kernel_size, stride, True, False, *args, **kwargs)

    def import_parameters(self, scale_in, zero_point_in, conv, prelu):
        super().import_parameters(scale_in, zero_point_in, None, None, conv, prelu)


class PReLUIn32Out32(nn.Module):
    def __init__(self):
        super().__init__()
        self.register_buffer('slope', torch.zeros((1,), dtype=torch.int32), persistent=True)

    @torch.no_grad()
    def import_parameters(self, prelu):
        self.slope[:] = (prelu.weight.detach().float().reshape(1) * (1 << 25)).round().to(torch.int32)   # Q6.25

    def forward(self, input: torch.Tensor, add: Optional[torch.Tensor] = None) -> torch.Tensor:
        return ops.prelu_i32(input, self.slope, add)


class RequantFxpToScaledInt8(LoadSaveUint32RequantMul):
    """Q8.23 -> per-tensor scaled int8 (cuda_ops.py:473-509)"""

    def __init__(self, eps=None, requant_mul_guard_bits=None):
        super().__init__()
        self.register_buffer('requant_mul', torch.zeros((1,), dtype=torch.uint32), persistent=True)
        self.register_buffer('requant_shift', torch.zeros((1,), dtype=torch.int32), persistent=True)
        self.register_buffer('int_zero_point_out', torch.zeros((1,), dtype=torch.int64), persistent=True)
        self.register_buffer('scale_out', torch.zeros((1,), dtype=torch.float32) - 1, persistent=True)
        self.register_buffer('zero_point_out', torch.zeros((1,), dtype=torch.float32), persistent=True)
        self.eps = eps if eps is not None else torch.finfo(torch.float32).eps
        self.requant_mul_guard_bits = requant_mul_guard_bits if requant_mul_guard_bits is not None else 2
        self._shift_host: Optional[int] = None

    @torch.no_grad()
    def import_parameters(self, scale_out: torch.Tensor, zero_point_out: torch.Tensor):
        scale_out, zero_point_out = scale_out.to(self.scale_out.device), zero_point_out.to(self.scale_out.device)
        scale_out = scale_out.reshape(1).float().clip(min=self.eps)
        self.scale_out[:] = scale_out
        shift = torch.log2((1 << (32 - self.requant_mul_guard_bits)) * scale_out).floor()
        if shift < 0:
            raise ValueError(f'negative requant shift {shift}')
        self.requant_shift[:] = shift.to(torch.int32)
        self.requant_mul[:] = ((2 ** shift.to(torch.float)) / scale_out).round().to(torch.int64).to(torch.uint32)
        self.zero_point_out[:] = zero_point_out.reshape(1)
        self.int_zero_point_out[:] = (zero_point_out.reshape(1) * (2 ** (SharedFxpShift + shift.to(torch.float)))).round().to(torch.int64)
        self._shift_host = None

    def _load_from_state_dict(self, *args, **kwargs):
        self._shift_host = None
        super()._load_from_state_dict(*args, **kwargs)

    def hint(self, width: int):
        """(multiplier, zero point, shift, row width) for a producer that writes this module's output from its own epilogue"""
        if self._shift_host is None:
            self._shift_host = int(self.requant_shift.item())
        return self.requant_mul, self.int_zero_point_out, SharedFxpShift + self._shift_host, width

    def forward(self, input: torch.Tensor) -> torch.Tensor:
        ready = getattr(input, '_fpcc_q8', None)
        if ready is not None and id(self) in ready:          # written by the producer of `input` (see `_with_q8`)
            return ready[id(self)]
        if self._shift_host is None:
            self._shift_host = int(self.requant_shift.item())
        return ops.epilogue_i32(input, self.requant_mul, self.int_zero_point_out, SharedFxpShift + self._shift_host, 8)


class SparseResBlockIn32W8Out32(nn.Module):
    def __init__(self, ch: int, eps=None):
        super().__init__()
        self.ch = ch
        self.eps = eps if eps is not None else torch.finfo(torch.float32).eps
        self.input_requant = RequantFxpToScaledInt8()
        self.conv_prelu = SparseConvPReLUIn8W8Out8(ch, ch, (3, 3, 3), (1, 1, 1))
        self.conv2 = SparseConvIn8W8Out32(ch, ch, (3, 3, 3), (1, 1, 1))
        self.prelu = PReLUIn32Out32()

    @torch.no_grad()
    def import_parameters(self, block):
        """block: calibrated float residual block (obs, conv, act, obs2, conv2, act2), cuda_ops.py:65-77"""
        scale, zero_point = block.obs.calculate_qparams()
        scale2, zero_point2 = block.obs2.calculate_qparams()
        self.input_requant.import_parameters(scale, zero_point)
        self.conv_prelu.import_parameters(scale, zero_point, scale2, zero_point2, block.conv, block.act)    # int8 -> int8
        self.conv2.import_parameters(scale2, zero_point2, block.conv2)                                      # int8 -> Q8.23
        self.prelu.import_parameters(block.act2)

    def forward(self, input: SparseTensor, _also=None) -> SparseTensor:
        """_also: requantisers of this block's consumers; their int8 outputs are written by conv2's epilogue"""
        x = SparseTensor(self.input_requant(input.F), input.C, input.stride, input.spatial_range)
        x._caches = input._caches
        # prelu(input + conv2(...)) runs in the epilogue of conv2 (fpcc_conv_i8_res): three launches per block instead of four
        # and the [N, C] int32 intermediate never reaches HBM
        res = input.F if input.F.is_contiguous() else input.F.contiguous()
        x = self.conv2.forward_with_sparse_tensor(self.conv_prelu(x), _residual=(res, self.prelu.slope), _also=_also)
        out = SparseTensor(x.F, input.C, input.stride, input.spatial_range)
        out._caches = input._caches
        return out


class LinearIn8W8(_RequantParams):
    def __init__(self, in_ch: int, out_ch: int, with_prelu: bool, out_scaled_int: bool, eps=None, requant_mul_guard_bits=None):
        super().__init__()
        self._make_buffers((out_ch, in_ch), out_ch, with_prelu)
        self.in_ch, self.out_ch = in_ch, out_ch
        self.with_prelu, self.out_scaled_int = with_prelu, out_scaled_int
        self.eps = eps if eps is not None else torch.finfo(torch.float32).eps
        self.requant_mul_guard_bits = requant_mul_guard_bits if requant_mul_guard_bits is not None else 7

    @torch.no_grad()
    def import_parameters(self, scale_in, zero_point_in, scale_out, zero_point_out, linear, prelu):
        """linear: object with .weight [out, in] float and .bias"""
        self._import_common(scale_in, zero_point_in, scale_out, zero_point_out, linear.weight.detach().float(),
                            linear.bias.detach().float().reshape(-1), None if prelu is None else prelu.weight.detach().float(),
                            fold_zero_point_into_bias=True)

    def forward(self, input: torch.Tensor, _also=None) -> torch.Tensor:
        ep = self._epilogue()
        if _also and not self.out_scaled_int:
            hints, ep['also'] = _also_of(_also, self.out_ch)
            out, extra = ops.conv_i8(input, self._padded_weight(), self.in_ch, self.out_ch, input.shape[0], **ep)
            return _with_q8(out, hints, extra)
        return ops.conv_i8(input, self._padded_weight(), self.in_ch, self.out_ch, input.shape[0], **ep)


class LinearIn8W8Out8(LinearIn8W8):
    def __init__(self, in_ch, out_ch, *args, **kwargs):
        super().__init__(in_ch, out_ch, False, True, *args, **kwargs)

    def import_parameters(self, scale_in, zero_point_in, scale_out, zero_point_out, linear):
        super().import_parameters(scale_in, zero_point_in, scale_out, zero_point_out, linear, None)


class LinearIn8W8Out32(LinearIn8W8):
    def __init__(self, in_ch, out_ch, *args, **kwargs):
        super().__init__(in_ch, out_ch, False, False, *args, **kwargs)

    def import_parameters(self, scale_in, zero_point_in, linear):
        super().import_parameters(scale_in, zero_point_in, None, None, linear, None)


class LinearPReLUIn8W8Out8(LinearIn8W8):
    def __init__(self, in_ch, out_ch, *args, **kwargs):
        super().__init__(in_ch, out_ch, True, True, *args, **kwargs)

    def import_parameters(self, scale_in, zero_point_in, scale_out, zero_point_out, linear, prelu):
        super().import_parameters(scale_in, zero_point_in, scale_out, zero_point_out, linear, prelu)


class LinearPReLUIn8W8Out32(LinearIn8W8):
    def __init__(self, in_ch, out_ch, *args, **kwargs):
        super().__init__(in_ch, out_ch, True, False, *args, **kwargs)

    def import_parameters(self, scale_in, zero_point_in, linear, prelu):
        super().import_parameters(scale_in, zero_point_in, None, None, linear, prelu)
