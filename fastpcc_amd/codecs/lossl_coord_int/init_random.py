"""Seeded random parameters for the integer codec (no checkpoint is available): float kernels are drawn and pushed
through the reference's float -> fixed-point conversion (`import_parameters`, cuda_ops.py:223-301,488-503,542-607) with one
common activation scale, so every tensor of the pipeline is populated with self-consistent values."""
from types import SimpleNamespace

import torch
import torch.nn as nn

from ...int_sparse_conv import LinearIn8W8, PReLUIn32Out32, RequantFxpToScaledInt8, SparseConvIn8Out8

ACT_SCALE = 4.0 / 127.0      # real value of one int8 step, shared by every scaled-int tensor


@torch.no_grad()
def randomize_(model: nn.Module, seed: int = 0, gain: float = 1.0) -> nn.Module:
    g = torch.Generator().manual_seed(seed)
    scale = torch.tensor([ACT_SCALE])
    zero = torch.zeros(1, dtype=torch.int64)

    def uni(shape, bound):
        return (torch.rand(shape, generator=g) * 2 - 1) * bound

    for name, m in model.named_modules():
        dev = next(m.buffers()).device if any(True for _ in m.buffers()) else 'cpu'
        if isinstance(m, RequantFxpToScaledInt8):
            m.import_parameters(scale.to(dev), zero.to(dev))
        elif isinstance(m, PReLUIn32Out32):
            m.import_parameters(SimpleNamespace(weight=torch.tensor([0.1 + 0.3 * torch.rand(1, generator=g).item()]).to(dev)))
        elif isinstance(m, SparseConvIn8Out8):
            fan = m.in_ch * max(1.0, m.kernel_volume / 4.0)          # a few neighbours exist on LiDAR sweeps
            conv = SimpleNamespace(kernel=uni((m.kernel_volume, m.in_ch, m.out_ch), gain * (3.0 / fan) ** 0.5).to(dev),
                                   bias=uni((m.out_ch,), 0.2).to(dev))
            prelu = SimpleNamespace(weight=torch.tensor([0.25]).to(dev)) if m.with_prelu else None
            out = (scale.to(dev), zero.to(dev)) if m.out_scaled_int else (None, None)
            SparseConvIn8Out8.import_parameters(m, scale.to(dev), zero.to(dev), out[0], out[1], conv, prelu)
        elif isinstance(m, LinearIn8W8):
            lin = SimpleNamespace(weight=uni((m.out_ch, m.in_ch), gain * (3.0 / m.in_ch) ** 0.5).to(dev),
                                  bias=uni((m.out_ch,), 0.2).to(dev))
            prelu = SimpleNamespace(weight=torch.tensor([0.25]).to(dev)) if m.with_prelu else None
            out = (scale.to(dev), zero.to(dev)) if m.out_scaled_int else (None, None)
            LinearIn8W8.import_parameters(m, scale.to(dev), zero.to(dev), out[0], out[1], lin, prelu)
    return model
