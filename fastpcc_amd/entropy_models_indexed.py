"""Index-conditioned entropy models -- the Gaussian-conditional family of the reference:

    ContinuousIndexedEntropyModel                       /root/reference/lib/entropy_models/continuous_indexed.py:16-262
    noisy_scale_normal_indexed_entropy_model_init       continuous_indexed.py:265-273
    Normal, NoisyNormal, UniformNoiseAdapter            distributions/uniform_noise.py:20-116
    ndtr, log_ndtr                                      distributions/special_math.py:93-258

Every coded value carries an index (or a vector of indexes) that selects its prior from a grid of
`prod(index_ranges)` distributions; the grid's quantised CDF tables live in the range coder
(`IndexedRansCoder.encode_with_indexes`, libfpcc_host).  Like the reference's, the module is host-side PyTorch glue
around the coder.  What is native here is the training rate term of the scale-indexed noisy normal
(`ScaleNoisyNormal`): log-probabilities, their sum and both gradients come from one HIP kernel
(fpcc_noisy_normal_bits_f32) instead of ~40 tensor ops; `log_prob` keeps the tensor-op formulation (CPU, tests, other
priors).  Same constructor arguments, method names and state-dict layout (`prior._extra_state`) as the reference."""
import math
from typing import Any, Callable, Dict, List, Tuple, Union

import numpy as np
import torch
import torch.nn as nn

from .entropy_models import DistributionQuantizedCDFTable, grad_scaler, lower_bound, upper_bound
from .sparse_conv_layers import minkowski_tensor_wrapped_fn

_LOWER32, _UPPER32, _LOWER64, _UPPER64 = -10.0, 5.0, -20.0, 8.0
_HALF_LOG_2PI = 0.5 * math.log(2.0 * math.pi)


def ndtr(x: torch.Tensor) -> torch.Tensor:
    """standard normal CDF, erf in the centre and erfc in the tails (special_math.py:125-135)"""
    h = 0.5 * math.sqrt(2.0)
    w = x * h
    z = w.abs()
    return 0.5 * torch.where(z < h, 1.0 + torch.erf(w), torch.where(w > 0.0, 2.0 - torch.erfc(z), torch.erfc(z)))


def log_ndtr(x: torch.Tensor, series_order: int = 3) -> torch.Tensor:
    """log of the standard normal CDF in three segments (special_math.py:138-258): x > upper: -ndtr(-x);
    x < lower: asymptotic series; else log(ndtr(x)).  The unchosen branches are fed clamped arguments so that their
    gradients stay finite."""
    lower, upper = (_LOWER64, _UPPER64) if x.dtype == torch.float64 else (_LOWER32, _UPPER32)
    lo = torch.full((), lower, dtype=x.dtype, device=x.device)
    xl = torch.minimum(x, lo)
    x2 = xl * xl
    even, odd, x2n = torch.zeros_like(x), torch.zeros_like(x), x2
    for n in range(1, series_order + 1):
        term = float(np.prod(np.arange(2 * n - 1, 1, -2))) / x2n           # (2n-1)!! / x^(2n)
        if n % 2:
            odd = odd + term
        else:
            even = even + term
        x2n = x2n * x2
    tail = -0.5 * x2 - torch.log(-xl) - _HALF_LOG_2PI + torch.log(1.0 + even - odd)
    return torch.where(x > upper, -ndtr(-x), torch.where(x > lo, torch.log(ndtr(torch.maximum(x, lo))), tail))


class Normal:
    """location-scale normal with log-CDF / log-survival functions (uniform_noise.py:96-104)"""

    def __init__(self, loc, scale):
        self.loc, self.scale = loc, scale
        shape = torch.broadcast_shapes(*(t.shape for t in (loc, scale) if isinstance(t, torch.Tensor)))
        self.batch_shape, self.event_shape = torch.Size(shape), torch.Size([])

    def _z(self, x):
        return (x - self.loc) / self.scale

    def log_cdf(self, x):
        return log_ndtr(self._z(x))

    def log_survival_function(self, x):
        return log_ndtr(-self._z(x))

    def cdf(self, x):
        # operation order of torch.distributions.Normal.cdf (the difference of two of these is taken: cancellation)
        return 0.5 * (1 + torch.erf((x - self.loc) * torch.as_tensor(self.scale).reciprocal() / math.sqrt(2)))


class UniformNoiseAdapter:
    """density of X + U(-h, h): cdf(y + h) - cdf(y - h), evaluated with survival functions right of the median when the
    base distribution has them (uniform_noise.py:20-87)"""

    def __init__(self, base, noise_width: float = 1):
        self.base, self.half_width = base, noise_width / 2
        self.batch_shape, self.event_shape = base.batch_shape, torch.Size([])

    def log_prob(self, y):
        hi, lo = y + self.half_width, y - self.half_width
        logcdf_hi, logcdf_lo = self.base.log_cdf(hi), self.base.log_cdf(lo)
        if hasattr(self.base, 'log_survival_function'):
            logsf_hi, logsf_lo = self.base.log_survival_function(hi), self.base.log_survival_function(lo)
            right = logsf_hi < logcdf_hi
            big = torch.where(right, logsf_lo, logcdf_hi)
            small = torch.where(right, logsf_hi, logcdf_lo)
        else:
            big, small = logcdf_hi, logcdf_lo
        return torch.log1p(-torch.exp(small - big)) + big

    def prob(self, y):
        hi, lo = y + self.half_width, y - self.half_width
        if hasattr(self.base, 'survival_function'):
            sf_hi, sf_lo, cdf_hi, cdf_lo = self.base.survival_function(hi), self.base.survival_function(lo), self.base.cdf(hi), self.base.cdf(lo)
            return torch.where(sf_hi < cdf_hi, sf_lo - sf_hi, cdf_hi - cdf_lo)
        return self.base.cdf(hi) - self.base.cdf(lo)


class NoisyNormal(UniformNoiseAdapter):
    def __init__(self, loc, scale):
        super().__init__(Normal(loc, scale))


def noisy_scale_normal_indexed_entropy_model_init(scale_min: float, scale_max: float, num_scales: int) -> Dict[str, Callable]:
    """index i -> zero-mean normal of scale exp(log scale_min + i (log scale_max - log scale_min) / (num_scales - 1))"""
    offset = math.log(scale_min)
    factor = (math.log(scale_max) - math.log(scale_min)) / (num_scales - 1)
    fns = {'loc': lambda _: 0, 'scale': lambda i: torch.exp(offset + factor * i)}
    fns['scale'].log_scale_affine = (offset, factor)          # what the fused rate kernel needs to know
    return fns


class _NoisyNormalBits(torch.autograd.Function):
    """sum of log-probabilities of y under N(0, exp(a + b i)) + U(-.5, .5), with d/dy and d/di, in one kernel"""

    @staticmethod
    def forward(ctx, y, index, a, b):
        from . import hipops as ops
        total, dy, di = ops.noisy_normal_bits(y.contiguous().view(-1), index.contiguous().view(-1), a, b)
        ctx.save_for_backward(dy, di)
        ctx.shapes = (y.shape, index.shape)
        return total

    @staticmethod
    def backward(ctx, g):
        dy, di = ctx.saved_tensors
        return (dy * g).view(ctx.shapes[0]), (di * g).view(ctx.shapes[1]), None, None


class ContinuousIndexedEntropyModel(nn.Module):
    """prior_fn(**{name: fn(indexes)}) builds the prior; `indexes` has the bottleneck's shape (plus one innermost axis when
    there are several index channels), channel k in [0, index_ranges[k]).
    training: (x + U(-.5,.5), {'bits_loss'}); eval: (decoded x, [bytes])."""

    def __init__(self, prior_fn: Callable, index_ranges: Tuple[int, ...], parameter_fns: Dict[str, Callable[[torch.Tensor], Any]],
                 coding_ndim: int, bottleneck_process: str = 'noise', bottleneck_scaler: int = 1,
                 quantize_bottleneck_in_eval: bool = True, indexes_bound_gradient: str = 'identity_if_towards',
                 quantize_indexes: bool = False, indexes_scaler: float = 1, indexes_offset: float = 0,
                 lower_bound: Union[int, torch.Tensor] = -64, upper_bound: Union[int, torch.Tensor] = 64,
                 batch_shape: torch.Size = torch.Size([1]), overflow_coding: bool = True):
        super().__init__()
        self.additional_indexes_dim = len(index_ranges) != 1
        self.prior_fn, self.parameter_fns = prior_fn, parameter_fns
        self.bottleneck_scaler = bottleneck_scaler
        self.quantize_bottleneck_in_eval = quantize_bottleneck_in_eval
        self.index_ranges = tuple(index_ranges)
        self.indexes_bound_gradient = indexes_bound_gradient
        self.quantize_indexes = quantize_indexes
        self.indexes_scaler, self.indexes_offset = indexes_scaler, indexes_offset
        grid = self.make_range_coding_prior_indexes()
        with torch.no_grad():
            prior = self.make_prior(grid)
        self.prior = DistributionQuantizedCDFTable(prior, lower_bound, upper_bound, torch.Size(batch_shape).numel(),
                                                   overflow_coding, bottleneck_scaler)
        proc = bottleneck_process
        self.quantize_bottleneck = 'quantization' in proc
        proc = proc.replace('quantization', '', 1)
        self.perturb_bottleneck = 'noise' in proc
        proc = proc.replace('noise', '', 1)
        if proc not in (',', '_', ' ', '+', ''):
            raise ValueError(f'Unexpected bottleneck_process: {bottleneck_process}')
        self.coding_ndim = coding_ndim
        self.register_buffer('range_coding_prior_indexes', grid, persistent=False)

    # -- prior ---------------------------------------------------------------------------------------------------------
    def _parameters_of(self, indexes: torch.Tensor) -> torch.Tensor:
        if indexes.requires_grad:
            if not self.training:
                raise RuntimeError('differentiable indexes outside training')
            if self.quantize_indexes:
                indexes = indexes + (indexes.detach().round() - indexes.detach())
        else:
            indexes = indexes.round()
        if self.indexes_scaler != 0:
            indexes = indexes / self.indexes_scaler
        else:
            span = torch.tensor([r - 1 for r in self.index_ranges], dtype=indexes.dtype, device=indexes.device)
            indexes = (indexes / span - 0.5) * 2
        return indexes - self.indexes_offset

    def make_prior(self, indexes: torch.Tensor):
        i = self._parameters_of(indexes)
        return self.prior_fn(**{k: f(i) for k, f in self.parameter_fns.items()})

    @torch.no_grad()
    def update_prior(self):
        self.prior.update_base(self.make_prior(self.range_coding_prior_indexes))

    def make_range_coding_prior_indexes(self) -> torch.Tensor:
        if not self.additional_indexes_dim:
            return torch.arange(self.index_ranges[0]).to(torch.float)
        axes = torch.meshgrid(*[torch.arange(r) for r in self.index_ranges], indexing='ij')
        return torch.stack(axes, dim=-1).to(torch.float)

    def bound_indexes(self, indexes: torch.Tensor) -> torch.Tensor:
        indexes = indexes + self.indexes_offset
        if self.indexes_scaler != 0:
            indexes = indexes * self.indexes_scaler
        else:
            span = torch.tensor([r - 1 for r in self.index_ranges], dtype=indexes.dtype, device=indexes.device)
            indexes = (indexes / 2 + 0.5) * span
        indexes = lower_bound(indexes, 0, self.indexes_bound_gradient)
        if not self.additional_indexes_dim:
            bounds = torch.tensor([self.index_ranges[0] - 1], dtype=torch.int32, device=indexes.device)
        else:
            bounds = torch.tensor([r - 1 for r in self.index_ranges], dtype=torch.int32, device=indexes.device)
            bounds = bounds.reshape([1] * (indexes.ndim - 1) + [len(self.index_ranges)])
        return upper_bound(indexes, bounds, self.indexes_bound_gradient)

    @torch.no_grad()
    def flatten_indexes(self, indexes: torch.Tensor) -> torch.Tensor:
        indexes = indexes.round()
        if not self.additional_indexes_dim:
            return indexes.to(torch.int32)
        strides = torch.cumprod(torch.tensor((1, *self.index_ranges[:0:-1]), device=indexes.device, dtype=torch.float), dim=0)
        return torch.tensordot(indexes, torch.flip(strides, dims=[0]), [[-1], [0]]).to(torch.int32)

    # -- bottleneck processing (continuous_base.py:171-197) ----------------------------------------------------------------
    def process(self, x: torch.Tensor) -> torch.Tensor:
        if self.quantize_bottleneck:
            x = x + (x.detach().round() - x.detach())
        if self.perturb_bottleneck:
            x = x + torch.empty_like(x).uniform_(-0.5, 0.5)
        return x

    def _fused_rate(self, y: torch.Tensor, bounded: torch.Tensor):
        """the one-kernel rate term, when the prior is the scale-indexed noisy normal on the GPU; None otherwise"""
        affine = getattr(self.parameter_fns.get('scale'), 'log_scale_affine', None)
        if affine is None or self.prior_fn is not NoisyNormal or self.additional_indexes_dim or not y.is_cuda or \
                y.dtype != torch.float32 or self.quantize_indexes or self.indexes_scaler == 0 or bounded.shape != y.shape:
            return None
        a, b = affine
        # scale = exp(a + b (i / scaler - offset))
        return _NoisyNormalBits.apply(y, bounded, a - b * self.indexes_offset, b / self.indexes_scaler)

    @minkowski_tensor_wrapped_fn({1: 0, 2: None})
    def forward(self, x: torch.Tensor, indexes: torch.Tensor, is_first_forward: bool = True,
                x_grad_scaler_for_bits_loss: float = 1.0):
        if self.bottleneck_scaler != 1:
            x = x * self.bottleneck_scaler
        if self.training:
            indexes = self.bound_indexes(indexes)
            if is_first_forward:
                self.update_prior()
            y = self.process(x)
            if self.bottleneck_scaler != 1:
                y = y / self.bottleneck_scaler
            scaled = grad_scaler(y, x_grad_scaler_for_bits_loss)
            total = self._fused_rate(scaled, indexes)
            if total is None:
                total = self.make_prior(indexes).log_prob(scaled).sum()
            return y, {'bits_loss': total / (-math.log(2))}
        bytes_list, _ = self.compress(x, indexes)
        return self.decompress(bytes_list, indexes, x.device), bytes_list

    @torch.no_grad()
    @minkowski_tensor_wrapped_fn({1: 1, 2: None})
    def compress(self, x: torch.Tensor, indexes: torch.Tensor, estimate_bits: bool = False):
        if self.bottleneck_scaler != 1:
            x = x * self.bottleneck_scaler
        unit = x.shape[-self.coding_ndim:]
        indexes = self.bound_indexes(indexes)
        flat = self.flatten_indexes(indexes)
        if flat.shape != x.shape:
            raise ValueError(f'indexes of shape {tuple(indexes.shape)} do not match the bottleneck {tuple(x.shape)}')
        flat = flat.reshape(-1, unit.numel())
        if self.quantize_bottleneck_in_eval:
            deq = x.round()
            q = deq.to(torch.int32)
        else:
            deq, q = x, x.to(torch.int32)
        strings = self.prior.range_coder.encode_with_indexes(q.reshape(-1, unit.numel()).cpu().numpy(), flat.cpu().numpy())
        if self.bottleneck_scaler != 1:
            deq = deq / self.bottleneck_scaler
        if estimate_bits:
            return strings, deq, self.make_prior(indexes).log_prob(q / self.bottleneck_scaler).sum() / (-math.log(2))
        return strings, deq

    @torch.no_grad()
    @minkowski_tensor_wrapped_fn({'<del>sparse_tensor_coords_tuple': 0, 2: None})
    def decompress(self, bytes_list: List[bytes], indexes: torch.Tensor, target_device):
        flat = self.flatten_indexes(self.bound_indexes(indexes))
        shape = flat.shape
        unit = shape[-self.coding_ndim:]
        flat = flat.reshape(-1, unit.numel()).cpu().numpy()
        symbols = np.empty_like(flat)
        self.prior.range_coder.decode_with_indexes(bytes_list, flat, symbols)
        out = torch.from_numpy(symbols).to(target_device).to(torch.float).reshape(shape)
        if self.bottleneck_scaler != 1:
            out = out / self.bottleneck_scaler
        return out

    def train(self, mode: bool = True):
        if not mode:
            self.update_prior()              # eval(): refresh the grid's prior, then the table is rebuilt
        return super().train(mode)
