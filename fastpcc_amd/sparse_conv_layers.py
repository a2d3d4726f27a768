"""Layer blocks with the constructor signatures, attribute names (hence state_dict keys) and forward contract of
/root/reference/lib/minkowski_sparse_conv_layers.py:11-159,228-249 -- `get_act_module`, `MEMLPBlock`, `ConvBlock`,
`ConvTransBlock`, `GenConvTransBlock`, `NNSequentialWith*Args` -- on top of fastpcc_amd.engine.

Difference in execution only: convolution / linear, bias add and activation run as ONE kernel launch
(fpcc_conv_f32's fused epilogue) instead of three MinkowskiEngine ops.  With bn=True (no in-scope config) the block runs
unfused: convolution, MinkowskiBatchNorm, activation.
"""
import functools
import math
from typing import Any, Callable, Dict, List, Optional, Tuple, Union

import torch
from torch import nn

from . import engine as ME


def get_act_module(act: Union[str, nn.Module, None]) -> Optional[nn.Module]:
    if isinstance(act, nn.Module):
        return act
    if act is None or act == 'None':
        return None
    if act == 'relu':
        return ME.MinkowskiReLU(inplace=True)
    if act.startswith('leaky_relu'):
        return ME.MinkowskiLeakyReLU(negative_slope=float(act.split('(', 1)[1].split(')', 1)[0]), inplace=True)
    if act == 'sigmoid':
        return ME.MinkowskiSigmoid()
    if act == 'prelu':
        return ME.MinkowskiPReLU()
    raise NotImplementedError(act)


def _fusable(module: Optional[nn.Module]) -> bool:
    return module is None or isinstance(module, (ME.MinkowskiPReLU, ME.MinkowskiReLU))


class MEMLPBlock(nn.Module):
    def __init__(self, in_channels: int, out_channels: int, bn: bool = False,
                 act: Union[str, nn.Module, None] = 'relu'):
        super().__init__()
        self.mlp = ME.MinkowskiLinear(in_channels, out_channels, bias=not bn)      # minkowski_sparse_conv_layers.py:38-39
        self.bn = ME.MinkowskiBatchNorm(out_channels) if bn else None
        self.act = get_act_module(act)

    def forward(self, x, clip: float = 0.0):
        if self.bn is None and _fusable(self.act):
            return self.mlp(x, act=ME._act_of(self.act), clip=clip)
        x = self.mlp(x)
        if self.bn is not None:
            x = self.bn(x)
        if self.act is not None:
            x = self.act(x)
        if clip > 0:
            x = ME.SparseTensor(x.F.clamp(-clip, clip), coordinate_map_key=x.coordinate_map_key, coordinate_manager=x.coordinate_manager)
        return x

    def __repr__(self):
        return f'MEMLPBlock(in_ch={self.mlp.linear.in_features}, out_ch={self.mlp.linear.out_features}, ' \
               f'bn={self.bn is not None}, act={self.act})'


# Stacks of MEMLPBlocks (per-point layers) run as ONE launch of fpcc_mlp_chain_f32 in inference when their shapes allow it: the
# activations between the layers stay in LDS.  Same bits as layer by layer (tests/test_gpu_mlp_chain.py); the switch exists
# for that test and for A/B timing.
FUSE_MLP_CHAINS = True
CHAIN_CALLS = 0


def mlp_chain_forward(blocks, x, y=None, cat_layer: int = -1, clip: float = 0.0):
    """`blocks`: MEMLPBlocks applied in sequence to x (SparseTensor or [n, C] tensor); `y` (SparseTensor) is concatenated
    after the activations entering block `cat_layer` (ME.cat(h, y)).  Returns the features [n, C_out] of the last block, or
    None when the stack cannot run fused (training / autograd, batch norm, an activation that does not fuse, shapes outside
    fpcc_mlp_chain_f32): the caller then evaluates block by block."""
    global CHAIN_CALLS
    from . import hipops as ops
    if not FUSE_MLP_CHAINS or torch.is_grad_enabled():
        return None
    xf = x if isinstance(x, torch.Tensor) else (x.parts[0] if len(x.parts) == 1 else None)
    yf = None
    if y is not None:
        if len(y.parts) != 1:
            return None
        yf = y.parts[0]
    if xf is None or xf.dim() != 2 or xf.dtype != torch.float32 or xf.stride(1) != 1:
        return None
    if any(b.bn is not None or not _fusable(b.act) for b in blocks):
        return None
    widths = [b.mlp.linear.out_features for b in blocks]
    if not ops.mlp_chain_ok(xf.shape[1], widths, cat_layer, 0 if yf is None else yf.shape[1]):
        return None
    layers = []
    for i, b in enumerate(blocks):
        act = ME._act_of(b.act)
        bias = b.mlp.linear.bias
        layers.append((b.mlp._weight_t(), None if bias is None else bias.detach(), act.kind, act.slope,
                       clip if i == len(blocks) - 1 else 0.0))
    CHAIN_CALLS += 1
    return ops.mlp_chain(xf, layers, y=yf, cat_layer=cat_layer)


class BaseConvBlock(nn.Module):
    def __init__(self, conv_class: Callable, in_channels, out_channels, kernel_size, stride, dilation=1, dimension=3,
                 region_type: str = 'HYPER_CUBE', bn: bool = False, bias: Optional[bool] = None,
                 act: Union[str, nn.Module, None] = 'relu'):
        super().__init__()
        self.region_type = getattr(ME.RegionType, region_type)
        self.conv = conv_class(
            in_channels, out_channels, kernel_size=kernel_size, stride=stride, dilation=dilation,
            bias=bias if bias is not None else not bn,                      # minkowski_sparse_conv_layers.py:67-81
            kernel_generator=ME.KernelGenerator(kernel_size, stride, dilation, region_type=self.region_type,
                                                dimension=dimension),
            dimension=dimension)
        self.bn = ME.MinkowskiBatchNorm(out_channels) if bn else None
        self.act = act
        self.act_module = get_act_module(act)

    def forward(self, x, *args, clip: float = 0.0, **kwargs):
        if self.bn is None and _fusable(self.act_module):
            return self.conv(x, *args, act=ME._act_of(self.act_module), clip=clip, **kwargs)
        x = self.conv(x, *args, **kwargs)
        if self.bn is not None:
            x = self.bn(x)
        if self.act_module is not None:
            x = self.act_module(x)
        if clip > 0:
            x = ME.SparseTensor(x.F.clamp(-clip, clip), coordinate_map_key=x.coordinate_map_key, coordinate_manager=x.coordinate_manager)
        return x

    def __repr__(self):
        kg = self.conv.kernel_generator
        return f'{type(self.conv).__name__}(in={self.conv.in_channels}, out={self.conv.out_channels}, ' \
               f'kernel_size={kg.kernel_size[0]}, stride={kg.kernel_stride[0]}, act={self.act})'


class ConvBlock(BaseConvBlock):
    def __init__(self, in_channels, out_channels, kernel_size, stride, dilation=1, dimension=3,
                 region_type: str = 'HYPER_CUBE', bn: bool = False, bias: Optional[bool] = None,
                 act: Union[str, nn.Module, None] = 'relu'):
        super().__init__(ME.MinkowskiConvolution, in_channels, out_channels, kernel_size, stride, dilation, dimension,
                         region_type, bn, bias, act)


class ConvTransBlock(BaseConvBlock):
    def __init__(self, in_channels, out_channels, kernel_size, stride, dilation=1, dimension=3,
                 region_type: str = 'HYPER_CUBE', bn: bool = False, bias: Optional[bool] = None,
                 act: Union[str, nn.Module, None] = 'relu'):
        super().__init__(ME.MinkowskiConvolutionTranspose, in_channels, out_channels, kernel_size, stride, dilation,
                         dimension, region_type, bn, bias, act)


class GenConvTransBlock(BaseConvBlock):
    def __init__(self, in_channels, out_channels, kernel_size, stride, dilation=1, dimension=3,
                 region_type: str = 'HYPER_CUBE', bn: bool = False, bias: Optional[bool] = None,
                 act: Union[str, nn.Module, None] = 'relu'):
        super().__init__(ME.MinkowskiGenerativeConvolutionTranspose, in_channels, out_channels, kernel_size, stride,
                         dilation, dimension, region_type, bn, bias, act)


class NNSequentialWithArgs(nn.Sequential):
    """Sequential that hands its extra arguments (a target coordinate key) to the first block of a given class."""
    target_block_class = None

    def forward(self, x, *args, **kwargs):
        handed = False
        for m in self:
            if not handed and isinstance(m, self.target_block_class):
                x = m(x, *args, **kwargs)
                handed = True
            else:
                x = m(x)
        if (args or kwargs) and not handed:
            raise RuntimeError('no block accepted the extra arguments')
        return x


class NNSequentialWithConvTransBlockArgs(NNSequentialWithArgs):
    target_block_class = ConvTransBlock


class NNSequentialWithConvBlockArgs(NNSequentialWithArgs):
    target_block_class = ConvBlock


# ---- sparse-tensor <-> plain-tensor adapters (lib/minkowski_sparse_conv_layers.py:252-400) -------------------------------------
def minkowski_tensor_wrapped_op(x, operation: Callable[[torch.Tensor], Any], needs_recover: bool = True,
                                add_batch_dim: bool = False):
    """apply `operation` to the feature matrix of a sparse tensor (or to a plain tensor as is); tensor results are wrapped
    back onto x's coordinate map when needs_recover, or get a leading batch dimension when add_batch_dim"""
    if needs_recover and add_batch_dim:
        raise ValueError('needs_recover and add_batch_dim exclude each other')
    if isinstance(x, torch.Tensor):
        return operation(x)
    ret = operation(x.F)
    items = list(ret) if isinstance(ret, tuple) else [ret]
    for i, r in enumerate(items):
        if isinstance(r, torch.Tensor):
            if needs_recover:
                items[i] = ME.SparseTensor(r, coordinate_map_key=x.coordinate_map_key, coordinate_manager=x.coordinate_manager)
            elif add_batch_dim:
                items[i] = r[None]
    return items[0] if len(items) == 1 else tuple(items)


def get_minkowski_tensor_coords_tuple(x):
    try:
        return x.coordinate_map_key, x.coordinate_manager
    except AttributeError:
        return None


def minkowski_tensor_wrapped_fn(inout_mapping_dict: Optional[Dict[Union[int, str], Union[int, List[int], None]]] = None,
                                add_batch_dim: bool = True):
    """Decorator for functions written on plain tensors.  `{arg: out}`: if argument `arg` (position or keyword) is a
    SparseTensor its features are passed instead ([1, N, C] when add_batch_dim) and result number `out` (or each of a
    list) is wrapped back onto its coordinate map; an argument may also be a (CoordinateMapKey, CoordinateManager) pair
    that only names the map; a key '<del>name' removes that argument before the call."""
    mapping = inout_mapping_dict or {}

    def decorate(func):
        @functools.wraps(func)
        def wrapped(*args, **kwargs):
            args = list(args)
            onto: Dict[int, Tuple] = {}
            for in_key, outs in mapping.items():
                drop = False
                if isinstance(in_key, str) and in_key.startswith('<'):
                    flag, in_key = in_key[1:].split('>', 1)
                    if flag != 'del':
                        raise NotImplementedError(flag)
                    drop = True
                if isinstance(in_key, str) and in_key.lstrip('-').isdigit():
                    in_key = int(in_key)
                bag = args if isinstance(in_key, int) else kwargs
                try:
                    obj = bag[in_key]
                except (IndexError, KeyError):
                    continue
                where = None
                if isinstance(obj, ME.SparseTensor):
                    bag[in_key] = obj.F[None] if add_batch_dim else obj.F
                    where = (obj.coordinate_map_key, obj.coordinate_manager)
                elif isinstance(obj, (tuple, list)) and len(obj) == 2 and isinstance(obj[0], ME.CoordinateMapKey):
                    where = tuple(obj)
                if where is not None and outs is not None:
                    for o in (outs if isinstance(outs, list) else [outs]):
                        onto[o] = where
                if drop:
                    del bag[in_key]
            ret = func(*args, **kwargs)
            if not onto:
                return ret
            items = list(ret) if isinstance(ret, tuple) else [ret]
            for o, (key, cm) in onto.items():
                f = items[o]
                items[o] = ME.SparseTensor(f[0] if add_batch_dim else f, coordinate_map_key=key, coordinate_manager=cm)
            return items[0] if len(items) == 1 else tuple(items)
        return wrapped
    return decorate


def minkowski_tensor_split(x, split_size: Union[int, List[int]]) -> List:
    """channel split of a sparse tensor: by a list of widths, or into blocks of `split_size` channels"""
    width = x.F.shape[1]
    if isinstance(split_size, list):
        ends = list(torch.cumsum(torch.tensor(split_size), 0).tolist())
    else:
        if math.ceil(width / split_size) < 2:
            raise ValueError('a split needs at least two blocks')
        ends = list(range(split_size, width, split_size)) + [width]
    starts = [0] + ends[:-1]
    return [ME.SparseTensor(x.F[:, a:b], coordinate_map_key=x.coordinate_map_key, coordinate_manager=x.coordinate_manager)
            for a, b in zip(starts, ends)]
