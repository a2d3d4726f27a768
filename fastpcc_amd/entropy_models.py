"""Continuous batched entropy model with a noisy deep-factorised prior: the `bottom_fea_entropy_model` of the codecs.

Same constructor arguments, attribute / state_dict names (`prior_weights.i`, `prior_biases.i`, `prior_factors.i`,
`prior._extra_state`) and numerics as the reference's PyTorch implementation:
    NoisyDeepFactorizedEntropyModel, ContinuousBatchedEntropyModel   /root/reference/lib/entropy_models/continuous_batched.py:17-200
    ContinuousEntropyModelBase, DistributionQuantizedCDFTable        /root/reference/lib/entropy_models/continuous_base.py:12-215
    DeepFactorized.logits_cdf / make_parameters                       /root/reference/lib/entropy_models/distributions/deep_factorized.py:24-77
    UniformNoiseAdapter (log_prob / prob with sf-cdf side selection)   /root/reference/lib/entropy_models/distributions/uniform_noise.py:12-87
    lower_bound / upper_bound / grad_scaler                            /root/reference/lib/entropy_models/utils.py:7-77
The reference implements these in plain PyTorch as well (they are host-side glue around the rANS coder, not kernels); this
is a restatement, not a copy: the density is evaluated by one function over stacked (y-h, y+h) inputs instead of a
Distribution class hierarchy.  Pinned against golden values produced by the reference (tests/golden/entropy_model.json).
"""
import io
import functools
import math
from typing import Dict, List, Optional, Tuple, Union

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .rans_coder import IndexedRansCoder
from .sparse_conv_layers import minkowski_tensor_wrapped_fn


# ---- gradient-shaping helpers --------------------------------------------------------------------------------------
class _Bound(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, bound, lower: bool):
        ctx.save_for_backward(x, bound)
        ctx.lower = lower
        return torch.max(x, bound) if lower else torch.min(x, bound)

    @staticmethod
    def backward(ctx, g):
        x, bound = ctx.saved_tensors
        # the gradient passes where the bound is inactive or where it pushes the value back inside
        ok = ((x >= bound) | (g < 0)) if ctx.lower else ((x <= bound) | (g > 0))
        return ok * g, None, None


@functools.lru_cache(maxsize=256)
def scalar_tensor(value: float, dtype: torch.dtype, device: torch.device) -> torch.Tensor:
    """a 1-element constant on `device`, made once: torch.tensor(..., device=cuda) is a host-to-device copy from pageable
    memory, which waits for the stream -- inside a training step that is a stall per call"""
    return torch.tensor([value], dtype=dtype, device=device)


def _as_bound(x, bound):
    return bound if isinstance(bound, torch.Tensor) else scalar_tensor(float(bound), x.dtype, x.device)


def lower_bound(x: torch.Tensor, bound, gradient: str = 'identity_if_towards') -> torch.Tensor:
    b = _as_bound(x, bound)
    if gradient == 'identity_if_towards':
        return _Bound.apply(x, b, True)
    if gradient == 'disconnected':
        return torch.maximum(x, b)
    raise NotImplementedError(gradient)


def upper_bound(x: torch.Tensor, bound, gradient: str = 'identity_if_towards') -> torch.Tensor:
    b = _as_bound(x, bound)
    if gradient == 'identity_if_towards':
        return _Bound.apply(x, b, False)
    if gradient == 'disconnected':
        return torch.minimum(x, b)
    raise NotImplementedError(gradient)


class _GradScale(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, factor):
        ctx.save_for_backward(factor)
        return x

    @staticmethod
    def backward(ctx, g):
        return g * ctx.saved_tensors[0], None


def grad_scaler(x: torch.Tensor, scaler) -> torch.Tensor:
    if not isinstance(scaler, torch.Tensor):
        if scaler == 1.0:
            return x
        scaler = scalar_tensor(float(scaler), x.dtype, x.device)
    return _GradScale.apply(x.clone(), scaler)


# ---- deep-factorised density ---------------------------------------------------------------------------------------
def make_parameters(batch_numel: int, init_scale: float = 10, num_filters: Tuple[int, ...] = (1, 3, 3, 3, 3, 1)):
    """weights [B, f_{i+1}, f_i] filled with softplus^-1(1 / scale / f_{i+1}); biases U(-.5, .5); factors 0"""
    if num_filters[0] != 1 or num_filters[-1] != 1:
        raise ValueError('the first and last filter counts must be 1')
    scale = init_scale ** (1 / (len(num_filters) + 1))
    weights, biases, factors = nn.ParameterList(), nn.ParameterList(), nn.ParameterList()
    for i in range(len(num_filters) - 1):
        fill = float(np.log(np.expm1(1 / scale / num_filters[i + 1])))
        weights.append(nn.Parameter(torch.full((batch_numel, num_filters[i + 1], num_filters[i]), fill)))
        biases.append(nn.Parameter(torch.empty((batch_numel, num_filters[i + 1], 1)).uniform_(-0.5, 0.5)))
        if i < len(num_filters) - 2:
            factors.append(nn.Parameter(torch.zeros((batch_numel, num_filters[i + 1], 1))))
    return weights, biases, factors


def logits_cdf(value: torch.Tensor, batch_shape: torch.Size, weights, biases, factors) -> torch.Tensor:
    """logit of the cumulative of every channel's learned density at `value` (trailing dims = batch_shape)"""
    shape = value.shape
    v = value.contiguous().view(-1, 1, batch_shape.numel()).permute(2, 1, 0).contiguous()      # [B, 1, n]
    last = len(weights) - 1
    for i, (w, b) in enumerate(zip(weights, biases)):
        v = torch.matmul(F.softplus(w), v) + b
        if i < last:
            v = v + torch.tanh(factors[i]) * torch.tanh(v)
    return v.permute(2, 1, 0).contiguous().view(shape)


class _NoisyDeepFactorized:
    """density of (X + U(-h, h)), X deep-factorised: p(y) = cdf(y + h) - cdf(y - h), evaluated on the numerically
    better side of the median"""

    def __init__(self, batch_shape: torch.Size, weights, biases, factors, noise_width: float):
        self.batch_shape, self.event_shape = batch_shape, torch.Size([])
        self.weights, self.biases, self.factors = weights, biases, factors
        self.half = noise_width / 2

    def _logits(self, y):
        return (logits_cdf(y + self.half, self.batch_shape, self.weights, self.biases, self.factors),
                logits_cdf(y - self.half, self.batch_shape, self.weights, self.biases, self.factors))

    def log_prob_sum(self, y: torch.Tensor) -> torch.Tensor:
        """sum of log_prob over all elements; on the GPU (and for the standard 1-3-3-3-3-1 network) one fused kernel"""
        if y.is_cuda and y.dtype == torch.float32 and len(self.weights) == 5 and \
                [tuple(w.shape[1:]) for w in self.weights] == [(3, 1), (3, 3), (3, 3), (3, 3), (1, 3)] and \
                y.shape[-len(self.batch_shape):] == self.batch_shape and len(self.batch_shape) == 1:
            return _DeepFactorizedBits.apply(y, self.half, *self.weights, *self.biases, *self.factors)
        return self.log_prob(y).sum()

    def log_prob(self, y: torch.Tensor) -> torch.Tensor:
        hi, lo = self._logits(y)
        right = F.logsigmoid(-hi) < F.logsigmoid(hi)            # right of the median: use survival functions
        big = torch.where(right, F.logsigmoid(-lo), F.logsigmoid(hi))
        small = torch.where(right, F.logsigmoid(-hi), F.logsigmoid(lo))
        return torch.log1p(-torch.exp(small - big)) + big

    def prob(self, y: torch.Tensor) -> torch.Tensor:
        hi, lo = self._logits(y)
        return torch.where(torch.sigmoid(-hi) < torch.sigmoid(hi), torch.sigmoid(-lo) - torch.sigmoid(-hi),
                           torch.sigmoid(hi) - torch.sigmoid(lo))


class _DeepFactorizedBits(torch.autograd.Function):
    """sum of log-probabilities of the noisy deep-factorised density as one device kernel (fpcc_deep_factorized_bits_f32):
    forward evaluates the sum AND every gradient in the same pass, backward scales them by the incoming gradient"""

    @staticmethod
    def forward(ctx, y, half, *params):
        from . import hipops as ops
        weights, biases, factors = params[0:5], params[5:10], params[10:14]
        c = y.shape[-1]
        flat = y.reshape(-1, c).contiguous()
        out, dy = ops.deep_factorized_bits(flat, weights, biases, factors, half, want_dy=True)
        ctx.save_for_backward(out, dy)
        ctx.y_shape = y.shape
        ctx.param_shapes = [p.shape for p in params]
        return out[:, 58].sum()

    @staticmethod
    def backward(ctx, g):
        out, dy = ctx.saved_tensors
        # the 14 parameter gradients: ONE scaling and ONE re-ordering launch (channel-major [c, 58] -> the parameters' own contiguous
        # blocks one after the other), then contiguous views of the result -- not 14 slice products (and no copies when autograd
        # stores them: a contiguous view obeys the parameter's layout)
        c = out.shape[0]
        flat = (out[:, :58] * g).reshape(-1)[_param_major(c, out.device)]
        grads, at = [], 0
        for shape, width in zip(ctx.param_shapes, _DF_WIDTHS):
            grads.append(flat[at * c: (at + width) * c].view(shape))
            at += width
        return (dy * g).reshape(ctx.y_shape), None, *grads


_DF_WIDTHS = (3, 9, 9, 9, 3, 3, 3, 3, 3, 1, 3, 3, 3, 3)      # columns of fpcc_deep_factorized_bits_f32's gradient rows, per parameter
_PARAM_MAJOR = {}


def _param_major(c: int, device) -> torch.Tensor:
    """index of element (parameter p, channel ch, j) in the flattened [c, 58] gradient matrix, parameters one after the other"""
    key = (c, str(device))
    perm = _PARAM_MAJOR.get(key)
    if perm is None:
        ch = torch.arange(c).view(c, 1)
        parts, at = [], 0
        for width in _DF_WIDTHS:
            parts.append((ch * 58 + at + torch.arange(width).view(1, width)).reshape(-1))
            at += width
        perm = _PARAM_MAJOR[key] = torch.cat(parts).to(device)
    return perm


# ---- quantised CDF table -------------------------------------------------------------------------------------------
class DistributionQuantizedCDFTable(nn.Module):
    """Caches the integer CDF table of a density for the range coder; rebuilt by .eval(), invalidated by .train(),
    carried in the state dict as `_extra_state`."""

    def __init__(self, base, lower_bound: int, upper_bound: int, coding_batch_size: int, overflow_coding: bool,
                 bottleneck_scaler: int):
        super().__init__()
        if not lower_bound < upper_bound:
            raise ValueError('lower_bound must be below upper_bound')
        self.base = base
        self.coding_batch_size, self.overflow_coding, self.bottleneck_scaler = coding_batch_size, overflow_coding, bottleneck_scaler
        self.register_buffer('lower_bound', torch.tensor(lower_bound, dtype=torch.int32).expand(base.batch_shape), persistent=False)
        self.register_buffer('upper_bound', torch.tensor(upper_bound, dtype=torch.int32).expand(base.batch_shape), persistent=False)
        self.cdf_list: List[List[int]] = [[]]
        self.cdf_offset_list = []
        self.requires_updating_cdf_table = True
        self.range_coder = IndexedRansCoder(overflow_coding, coding_batch_size)

    def update_base(self, new_base):
        """swap in a re-parameterised prior of the same shape (continuous_base.py:53-57)"""
        if type(new_base) is not type(self.base) or new_base.batch_shape != self.base.batch_shape:
            raise ValueError('the new prior must have the type and batch shape of the old one')
        self.base = new_base

    @property
    def batch_shape(self):
        return self.base.batch_shape

    @property
    def batch_ndim(self):
        return len(self.base.batch_shape)

    def log_prob(self, value):
        return self.base.log_prob(value)

    def prob(self, value):
        return self.base.prob(value)

    @torch.no_grad()
    def build_quantized_cdf_table(self):
        s = self.bottleneck_scaler
        minima = self.lower_bound * s
        length = int((self.upper_bound.max() * s).item()) - int(minima.max().item()) + 1
        grid = torch.arange(length, device=minima.device, dtype=torch.float)
        grid = grid.reshape(length, *[1] * self.batch_ndim) + minima[None, ...]
        pmf = self.prob(grid / s).reshape(length, -1)
        pmf = pmf.T.cpu().contiguous().numpy().astype(np.float64)
        offsets = minima.reshape(-1).cpu().contiguous().numpy().astype(np.int32)
        self.range_coder.init_with_pmfs(pmf, offsets)
        self.cdf_list = self.range_coder.get_cdfs()
        self.cdf_offset_list = self.range_coder.get_offset_array()
        self.requires_updating_cdf_table = False

    def get_extra_state(self):
        return self.cdf_list, self.cdf_offset_list, self.requires_updating_cdf_table

    def set_extra_state(self, state):
        if state[2]:
            print('Warning: cached cdf table in state dict requires updating; call model.eval() before inference.')
            return
        self.cdf_list, self.cdf_offset_list, self.requires_updating_cdf_table = state
        self.range_coder.init_with_quantized_cdfs(self.cdf_list, np.asarray(self.cdf_offset_list, dtype=np.int32))

    def train(self, mode: bool = True):
        if mode:
            self.requires_updating_cdf_table = True
        elif self.requires_updating_cdf_table:
            self.build_quantized_cdf_table()
        return super().train(mode)


# ---- the entropy model ---------------------------------------------------------------------------------------------
class NoisyDeepFactorizedEntropyModel(nn.Module):
    """x: [..., broadcast dims..., *batch_shape]; the innermost `coding_ndim` dims form one coded unit.
    The range coder is built for ONE coded unit per call (the reference's default batch_shape=[1]).
    training: returns (x + U(-.5,.5), {'bits_loss': -sum log2 p}); eval: (decoded x, [bytes], batch_shape)."""

    def __init__(self, batch_shape: torch.Size, coding_ndim: int, num_filters: Tuple[int, ...] = (1, 3, 3, 3, 3, 1),
                 bottleneck_process: str = 'noise', bottleneck_scaler: int = 1, quantize_bottleneck_in_eval: bool = True,
                 init_scale: float = 10, lower_bound: int = -64, upper_bound: int = 64, overflow_coding: bool = True,
                 broadcast_shape_bytes: Tuple[int, ...] = (2,)):
        super().__init__()
        batch_shape = torch.Size(batch_shape)
        weights, biases, factors = make_parameters(batch_shape.numel(), init_scale, num_filters)
        self.prior = DistributionQuantizedCDFTable(
            _NoisyDeepFactorized(batch_shape, weights, biases, factors, 1 / bottleneck_scaler),
            lower_bound, upper_bound, 1, overflow_coding, bottleneck_scaler)      # one coded unit per call, like the reference
        # registered after `prior`, as the reference does (continuous_batched.py:192-194): the state_dict lists
        # prior._extra_state before the parameter lists (tests/golden/me_semantics.json holds the reference's key order)
        self.prior_weights, self.prior_biases, self.prior_factors = weights, biases, factors
        proc = bottleneck_process
        self.quantize_bottleneck = 'quantization' in proc
        proc = proc.replace('quantization', '', 1)
        self.perturb_bottleneck = 'noise' in proc
        proc = proc.replace('noise', '', 1)
        if proc not in (',', '_', ' ', '+', ''):
            raise ValueError(f'Unexpected bottleneck_process: {bottleneck_process}')
        if coding_ndim < len(batch_shape):
            raise ValueError('coding_ndim must cover the prior batch dims')
        self.coding_ndim = coding_ndim
        self.bottleneck_scaler = bottleneck_scaler
        self.quantize_bottleneck_in_eval = quantize_bottleneck_in_eval
        self.broadcast_shape_bytes = tuple(broadcast_shape_bytes)
        self.prior_num_filter = num_filters

    def process(self, x: torch.Tensor) -> torch.Tensor:
        if self.quantize_bottleneck:
            x = x + (x.detach().round() - x.detach())
        if self.perturb_bottleneck:
            x = x + torch.empty_like(x).uniform_(-0.5, 0.5)
        return x

    @minkowski_tensor_wrapped_fn({1: 0})
    def forward(self, x: torch.Tensor):
        if self.bottleneck_scaler != 1:
            x = x * self.bottleneck_scaler
        if self.training:
            y = self.process(x)
            if self.bottleneck_scaler != 1:
                y = y / self.bottleneck_scaler
            return y, {'bits_loss': self.prior.base.log_prob_sum(y) / (-math.log(2))}
        bytes_list, batch_shape, _ = self.compress(x / self.bottleneck_scaler if self.bottleneck_scaler != 1 else x)
        return self.decompress(bytes_list, batch_shape, x.device), bytes_list, batch_shape

    @torch.no_grad()
    @minkowski_tensor_wrapped_fn({1: 2})
    def compress(self, x: torch.Tensor, estimate_bits: bool = False):
        s = self.bottleneck_scaler
        if s != 1:
            x = x * s
        unit = x.shape[-self.coding_ndim:]
        batch_shape = x.shape[:-self.coding_ndim]
        broadcast = unit[:len(unit) - self.prior.batch_ndim]
        if self.quantize_bottleneck_in_eval:
            x = x.round_()
        q = x.to(torch.int32)
        strings = self.prior.range_coder.encode(q.reshape(-1, unit.numel()).cpu().numpy())
        if len(self.broadcast_shape_bytes) != len(broadcast):
            raise ValueError('broadcast_shape_bytes does not match the tensor rank')
        if sum(self.broadcast_shape_bytes):
            head = b''.join(int(n).to_bytes(nb, 'little', signed=False) for nb, n in zip(self.broadcast_shape_bytes, broadcast))
            strings = [head + t for t in strings]
        deq = x / s if s != 1 else x
        if estimate_bits:
            return strings, batch_shape, deq, self.prior.log_prob(q / s).sum() / (-math.log(2))
        return strings, batch_shape, deq

    @torch.no_grad()
    @minkowski_tensor_wrapped_fn({'<del>sparse_tensor_coords_tuple': 0})
    def decompress(self, bytes_list: List[bytes], batch_shape: torch.Size, target_device, broadcast_shape=None):
        total = sum(self.broadcast_shape_bytes)
        if total:
            head, dims, at = bytes_list[0][:total], [], 0
            for nb in self.broadcast_shape_bytes:
                dims.append(int.from_bytes(head[at:at + nb], 'little', signed=False))
                at += nb
            broadcast_shape = torch.Size(dims)
            bytes_list = [t[total:] for t in bytes_list]
        else:
            broadcast_shape = torch.Size(broadcast_shape or [1] * len(self.broadcast_shape_bytes))
        batch_shape = torch.Size(batch_shape)
        sym = np.empty((batch_shape.numel(), broadcast_shape.numel() * self.prior.batch_shape.numel()), np.int32)
        self.prior.range_coder.decode(bytes_list, sym)
        out = torch.from_numpy(sym).to(target_device).to(torch.float)
        out = out.reshape(batch_shape + broadcast_shape + self.prior.batch_shape)
        if self.bottleneck_scaler != 1:
            out /= self.bottleneck_scaler
        return out

    def __repr__(self):
        return f'NoisyDeepFactorizedEntropyModel(batch_shape={tuple(self.prior.batch_shape)}, coding_ndim={self.coding_ndim}, ' \
               f'num_filter={self.prior_num_filter})'
