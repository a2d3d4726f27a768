"""Layer blocks with the constructor signatures, attribute names (hence state_dict keys) and forward contract of
/root/reference/lib/minkowski_sparse_conv_layers.py:11-159,228-249 -- `get_act_module`, `MEMLPBlock`, `ConvBlock`,
`ConvTransBlock`, `GenConvTransBlock`, `NNSequentialWith*Args` -- on top of fastpcc_amd.engine.

Difference in execution only: convolution / linear, bias add and activation run as ONE kernel launch
(fpcc_conv_f32's fused epilogue) instead of three MinkowskiEngine ops.  BatchNorm is not available (every in-scope config
uses bn=False).
"""
from typing import Callable, Optional, Union

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
        if bn:
            raise NotImplementedError('batch norm is not part of the in-scope configurations')
        self.mlp = ME.MinkowskiLinear(in_channels, out_channels, bias=True)
        self.bn = None
        self.act = get_act_module(act)

    def forward(self, x, clip: float = 0.0):
        if _fusable(self.act):
            return self.mlp(x, act=ME._act_of(self.act), clip=clip)
        x = self.act(self.mlp(x))
        return x

    def __repr__(self):
        return f'MEMLPBlock(in_ch={self.mlp.linear.in_features}, out_ch={self.mlp.linear.out_features}, ' \
               f'bn=False, act={self.act})'


class BaseConvBlock(nn.Module):
    def __init__(self, conv_class: Callable, in_channels, out_channels, kernel_size, stride, dilation=1, dimension=3,
                 region_type: str = 'HYPER_CUBE', bn: bool = False, bias: Optional[bool] = None,
                 act: Union[str, nn.Module, None] = 'relu'):
        super().__init__()
        if bn:
            raise NotImplementedError('batch norm is not part of the in-scope configurations')
        self.region_type = getattr(ME.RegionType, region_type)
        self.conv = conv_class(
            in_channels, out_channels, kernel_size=kernel_size, stride=stride, dilation=dilation,
            bias=bias if bias is not None else True,
            kernel_generator=ME.KernelGenerator(kernel_size, stride, dilation, region_type=self.region_type,
                                                dimension=dimension),
            dimension=dimension)
        self.bn = None
        self.act = act
        self.act_module = get_act_module(act)

    def forward(self, x, *args, clip: float = 0.0, **kwargs):
        if _fusable(self.act_module):
            return self.conv(x, *args, act=ME._act_of(self.act_module), clip=clip, **kwargs)
        return self.act_module(self.conv(x, *args, **kwargs))

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
