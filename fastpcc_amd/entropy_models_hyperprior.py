"""Hyperprior wrapper around the two entropy models (/root/reference/lib/entropy_models/hyperprior/noisy_deep_factorized/
basic.py:18-248): a hyper-encoder turns the latent y into side information z, coded with the batched deep-factorised model
(`hyperprior_entropy_model`); a hyper-decoder turns the decoded z into the indexes that select y's priors
(`prior_entropy_model`, the index-conditioned model).  Per coded unit the two byte strings are framed as
`len(prior) on prior_bytes_num_bytes little-endian bytes | prior | y`.  Same attribute names (hence state-dict keys),
constructor arguments and return conventions as the reference; `y` may be a tensor or a SparseTensor (batch size 1)."""
from functools import partial
from typing import Any, Callable, Dict, List, Tuple

import torch
import torch.nn as nn

from .entropy_models import NoisyDeepFactorizedEntropyModel as PriorEntropyModel, _NoisyDeepFactorized
from .entropy_models_indexed import ContinuousIndexedEntropyModel, NoisyNormal, noisy_scale_normal_indexed_entropy_model_init
from .sparse_conv_layers import get_minkowski_tensor_coords_tuple, minkowski_tensor_wrapped_op


def concat_loss_dicts(a: Dict[str, torch.Tensor], b: Dict[str, torch.Tensor], rename: Callable[[str], str] = lambda k: k):
    """merge b into a under renamed keys, adding where a key exists (lib/torch_utils.py:42-52)"""
    for k, v in b.items():
        k = rename(k)
        a[k] = a[k] + v if k in a else v
    return a


def noisy_deep_factorized_indexed_entropy_model_init(index_ranges: Tuple[int, ...], parameter_fns_type: str,
                                                     parameter_fns_factory: Callable[..., nn.Module], num_filters: Tuple[int, ...]):
    """Parameter functions of an index-conditioned deep-factorised prior (continuous_indexed.py:276-358): the index vector of a
    value either IS its network parameters ('split': one index channel per weight / bias / factor) or is mapped to them by
    small learned transforms ('transform').  -> (parameter_fns, indexes_view_fn, modules to register)"""
    if len(num_filters) < 2 or num_filters[0] != 1 or num_filters[-1] != 1:
        raise ValueError('num_filters must start and end with 1')
    if parameter_fns_type not in ('split', 'transform'):
        raise ValueError(parameter_fns_type)
    if len(index_ranges) < 2:
        raise NotImplementedError('a deep-factorised prior needs several index channels')
    channels = len(index_ranges)

    def indexes_view_fn(x):
        return minkowski_tensor_wrapped_op(x, lambda t: t.view(*t.shape[:-1], t.shape[-1] // channels, channels),
                                           needs_recover=False, add_batch_dim=True)

    n_w = [num_filters[i] * num_filters[i + 1] for i in range(len(num_filters) - 1)]
    n_b, n_f = list(num_filters[1:]), list(num_filters[1:-1])
    if parameter_fns_type == 'split':
        edges = torch.cumsum(torch.tensor([0, *n_w, *n_b, *n_f]), 0).tolist()
        if channels != edges[-1]:
            raise ValueError(f'{channels} index channels, {edges[-1]} network parameters')
        w_at, b_at, f_at = edges[:len(n_w) + 1], edges[len(n_w):len(n_w) + len(n_b) + 1], edges[len(n_w) + len(n_b):]
        fns = {
            'batch_shape': lambda i: i.shape[:-1],
            'weights': lambda i: [i[..., w_at[k]: w_at[k + 1]].reshape(-1, num_filters[k + 1], num_filters[k]) for k in range(len(n_w))],
            'biases': lambda i: [i[..., b_at[k]: b_at[k + 1]].reshape(-1, n_b[k], 1) for k in range(len(n_b))],
            'factors': lambda i: [i[..., f_at[k]: f_at[k + 1]].reshape(-1, n_f[k], 1) for k in range(len(n_f))],
        }
        return fns, indexes_view_fn, {}
    w_t = nn.ModuleList([parameter_fns_factory(channels, c) for c in n_w])
    b_t = nn.ModuleList([parameter_fns_factory(channels, c) for c in n_b])
    f_t = nn.ModuleList([parameter_fns_factory(channels, c) for c in n_f])
    fns = {
        'batch_shape': lambda i: i.shape[:-1],
        'weights': lambda i: [t(i).view(-1, num_filters[k + 1], num_filters[k]) for k, t in enumerate(w_t)],
        'biases': lambda i: [t(i).view(-1, n_b[k], 1) for k, t in enumerate(b_t)],
        'factors': lambda i: [t(i).view(-1, n_f[k], 1) for k, t in enumerate(f_t)],
    }
    return fns, indexes_view_fn, {'prior_indexes_weights_transforms': w_t, 'prior_indexes_biases_transforms': b_t,
                                  'prior_indexes_factors_transforms': f_t}


class EntropyModel(nn.Module):
    def __init__(self, hyper_encoder: nn.Module, hyper_decoder: nn.Module, hyperprior_batch_shape: torch.Size, coding_ndim: int,
                 prior_fn: Callable, index_ranges: Tuple[int, ...], parameter_fns: Dict[str, Callable[[torch.Tensor], Any]],
                 hyper_encoder_post_op: Callable = lambda x: x, hyper_decoder_post_op: Callable = lambda x: x,
                 hyperprior_num_filters: Tuple[int, ...] = (1, 3, 3, 3, 3, 1), hyperprior_init_scale: float = 10,
                 hyperprior_broadcast_shape_bytes: Tuple[int, ...] = (2,), prior_bytes_num_bytes: int = 2,
                 bottleneck_process: str = 'noise', bottleneck_scaler: int = 1,
                 indexes_bound_gradient: str = 'identity_if_towards', quantize_indexes: bool = False, indexes_scaler: float = 1):
        super().__init__()
        self.hyper_encoder, self.hyper_decoder = hyper_encoder, hyper_decoder
        self.hyper_encoder_post_op, self.hyper_decoder_post_op = hyper_encoder_post_op, hyper_decoder_post_op
        self.prior_bytes_num_bytes = prior_bytes_num_bytes
        self.hyperprior_entropy_model = PriorEntropyModel(
            batch_shape=hyperprior_batch_shape, coding_ndim=coding_ndim, num_filters=hyperprior_num_filters,
            bottleneck_process=bottleneck_process, bottleneck_scaler=bottleneck_scaler, init_scale=hyperprior_init_scale,
            broadcast_shape_bytes=hyperprior_broadcast_shape_bytes)
        self.prior_entropy_model = ContinuousIndexedEntropyModel(
            prior_fn=prior_fn, index_ranges=index_ranges, parameter_fns=parameter_fns, coding_ndim=coding_ndim,
            bottleneck_process=bottleneck_process, bottleneck_scaler=bottleneck_scaler,
            indexes_bound_gradient=indexes_bound_gradient, quantize_indexes=quantize_indexes, indexes_scaler=indexes_scaler)

    def forward(self, y, is_first_forward: bool = True):
        if self.training:
            z = self.hyper_encoder_post_op(self.hyper_encoder(y))
            z_tilde, hyper_loss = self.hyperprior_entropy_model(z)
            indexes = self.hyper_decoder_post_op(self.hyper_decoder(z_tilde))
            y_tilde, loss = self.prior_entropy_model(y, indexes, is_first_forward)
            return y_tilde, concat_loss_dicts(loss, hyper_loss, lambda k: 'hyper_' + k)
        strings, coding_batch_shape, _ = self.compress(y)
        where = get_minkowski_tensor_coords_tuple(y)
        return self.decompress(strings, coding_batch_shape, y.device, where), strings, coding_batch_shape

    def compress(self, y, estimate_bits: bool = False):
        z = self.hyper_encoder_post_op(self.hyper_encoder(y))
        prior_strings, coding_batch_shape, z_recon, *prior_bits = self.hyperprior_entropy_model.compress(z, estimate_bits=estimate_bits)
        indexes = self.hyper_decoder_post_op(self.hyper_decoder(z_recon))
        strings, deq_y, *bits = self.prior_entropy_model.compress(y, indexes, estimate_bits=estimate_bits)
        framed = self.concat_bytes_lists(prior_strings, strings)
        if bits:
            return framed, coding_batch_shape, deq_y, prior_bits[0] + bits[0]
        return framed, coding_batch_shape, deq_y

    def decompress(self, concat_bytes_list: List[bytes], coding_batch_shape: torch.Size, target_device,
                   sparse_tensor_coords_tuple: Tuple = None):
        prior_strings, strings = self.split_bytes_lists(concat_bytes_list)
        z_recon = self.hyperprior_entropy_model.decompress(prior_strings, coding_batch_shape, target_device,
                                                           sparse_tensor_coords_tuple=sparse_tensor_coords_tuple)
        pre_indexes = self.hyper_decoder(z_recon)
        where = get_minkowski_tensor_coords_tuple(pre_indexes)
        indexes = self.hyper_decoder_post_op(pre_indexes)
        return self.prior_entropy_model.decompress(strings, indexes, target_device, sparse_tensor_coords_tuple=where)

    def concat_bytes_lists(self, prior_bytes_list: List[bytes], bytes_list: List[bytes]) -> List[bytes]:
        return [len(p).to_bytes(self.prior_bytes_num_bytes, 'little', signed=False) + p + b for p, b in zip(prior_bytes_list, bytes_list)]

    def split_bytes_lists(self, concat_bytes_list: List[bytes]) -> Tuple[List[bytes], List[bytes]]:
        nb = self.prior_bytes_num_bytes
        priors, rest = [], []
        for s in concat_bytes_list:
            n = int.from_bytes(s[:nb], 'little', signed=False)
            if nb + n > len(s):
                raise ValueError('truncated string: the side-information length exceeds it')
            priors.append(s[nb: nb + n])
            rest.append(s[nb + n:])
        return priors, rest


class ScaleNoisyNormalEntropyModel(EntropyModel):
    """zero-mean normal whose scale index comes from the hyper-decoder; codes |y| (basic.py:158-203)"""

    def __init__(self, hyper_encoder: nn.Module, hyper_decoder: nn.Module, hyperprior_batch_shape: torch.Size, coding_ndim: int,
                 num_scales: int = 64, scale_min: float = 0.11, scale_max: float = 256,
                 hyperprior_num_filters: Tuple[int, ...] = (1, 3, 3, 3, 3, 1), hyperprior_init_scale: float = 10,
                 hyperprior_broadcast_shape_bytes: Tuple[int, ...] = (2,), prior_bytes_num_bytes: int = 2,
                 bottleneck_process: str = 'noise', indexes_bound_gradient: str = 'identity_if_towards',
                 quantize_indexes: bool = False, indexes_scaler: float = 1):
        super().__init__(hyper_encoder, hyper_decoder, hyperprior_batch_shape, coding_ndim, NoisyNormal, (num_scales,),
                         noisy_scale_normal_indexed_entropy_model_init(scale_min, scale_max, num_scales),
                         lambda x: x, lambda x: x, hyperprior_num_filters, hyperprior_init_scale,
                         hyperprior_broadcast_shape_bytes, prior_bytes_num_bytes, bottleneck_process, 1,
                         indexes_bound_gradient, quantize_indexes, indexes_scaler)

    def forward(self, y, is_first_forward: bool = True):
        return super().forward(minkowski_tensor_wrapped_op(y, torch.abs), is_first_forward)

    def compress(self, y, estimate_bits: bool = False):
        return super().compress(minkowski_tensor_wrapped_op(y, torch.abs), estimate_bits)


class NoisyDeepFactorizedEntropyModel(EntropyModel):
    """deep-factorised prior whose parameters are functions of a learned index vector (basic.py:205-248)"""

    def __init__(self, hyper_encoder: nn.Module, hyper_decoder: nn.Module, hyperprior_batch_shape: torch.Size, coding_ndim: int,
                 hyperprior_num_filters: Tuple[int, ...] = (1, 3, 3, 3, 3, 1), hyperprior_init_scale: float = 10,
                 hyperprior_broadcast_shape_bytes: Tuple[int, ...] = (2,), prior_bytes_num_bytes: int = 2,
                 index_ranges: Tuple[int, ...] = (16, 16, 16, 16), parameter_fns_type: str = 'transform',
                 parameter_fns_factory: Callable[..., nn.Module] = None, num_filters: Tuple[int, ...] = (1, 3, 3, 3, 1),
                 bottleneck_process: str = 'noise', bottleneck_scaler: int = 1,
                 indexes_bound_gradient: str = 'identity_if_towards', quantize_indexes: bool = False, indexes_scaler: float = 1):
        fns, view_fn, modules = noisy_deep_factorized_indexed_entropy_model_init(index_ranges, parameter_fns_type,
                                                                                 parameter_fns_factory, num_filters)
        EntropyModel.__init__(self, hyper_encoder, hyper_decoder, hyperprior_batch_shape, coding_ndim,
                              partial(_indexed_deep_factorized, noise_width=1 / bottleneck_scaler), index_ranges, fns,
                              lambda x: x, view_fn, hyperprior_num_filters, hyperprior_init_scale,
                              hyperprior_broadcast_shape_bytes, prior_bytes_num_bytes, bottleneck_process, bottleneck_scaler,
                              indexes_bound_gradient, quantize_indexes, indexes_scaler)
        for name, module in modules.items():              # the transforms the parameter functions close over
            setattr(self, name, module)

    def _apply(self, fn):
        super()._apply(fn)
        self.prior_entropy_model.update_prior()          # the grid prior holds tensors derived from moved parameters
        return self


def _indexed_deep_factorized(batch_shape, weights, biases, factors, noise_width: float = 1):
    return _NoisyDeepFactorized(torch.Size(batch_shape), weights, biases, factors, noise_width)
