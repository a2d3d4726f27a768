"""Back-propagation through the sparse convolutions (training path).

MinkowskiEngine provides this through its own autograd functions; the model code only ever calls `loss.backward()`
(/root/reference/train.py:262-270).  Here one torch.autograd.Function wraps the C ABI:

    forward   y  = fpcc_conv_f32(x, W)                                  (raw: bias / activation stay in autograd-land)
    backward  dx = fpcc_conv_f32(dy, W') on the MIRRORED row maps        -- the input gradient of a sparse convolution is
                   itself a sparse convolution:
                     3x3x3 on one map      : the 27-neighbour table is symmetric, W'[k] = W[26-k]^T
                     stride-2 2x2x2        : the transposed convolution over the same child_row table, W'[g] = W[g]^T
                     transposed / generative: the stride-2 convolution over the same table (a packed GEMM for a
                                              generated set), W'[g] = W[g]^T
                     per-point linear      : W' = W^T
              dW = fpcc_conv_wgrad_f32(x, dy) on the forward's row maps
so the backward pass runs on the same MFMA kernel as inference (plus the weight-gradient kernel).
"""
from typing import Optional

import torch

from . import hipops as ops


class ConvSpec:
    """row maps of one convolution call.  kind: 'k1' | 'k3' | 'k2s2' | 'k2s2T' | 'gen' | 'tab' (any lookup table
    [n_out][K] of input rows, -1 absent; weight gradient only -- used where the input carries no gradient)"""
    __slots__ = ('kind', 'n_in', 'n_out', 'table', 'row_order')

    def __init__(self, kind: str, n_in: int, n_out: int, table: Optional[torch.Tensor] = None,
                 row_order: Optional[torch.Tensor] = None):
        self.kind, self.n_in, self.n_out, self.table, self.row_order = kind, n_in, n_out, table, row_order


# Weights change every step, so a packed copy for the wave / persistent kernels would have to be made per call ('fresh': one small
# launch per convolution).  Measured on the cfg#5 step (tools/r02/train_ab.sh, back-to-back on one box): 59.5 / 57.3 ms packed against
# 57.8 / 58.9 ms on the workgroup-tiled kernel -- no gain, the step is bound by the weight-gradient kernels and launch count; off.
import os as _os
PACK = 'fresh' if _os.environ.get('FPCC_TRAIN_PACK', 'off') in ('fresh', '1') else False


def _strided_rows(t: torch.Tensor) -> bool:
    """a matrix the kernels read in place: unit column stride, rows at any pitch >= the row"""
    return t.dim() == 2 and t.stride(1) == 1 and (t.shape[0] <= 1 or t.stride(0) >= t.shape[1])


def _mfma(c_in: int, c_out: int) -> bool:
    return ops.conv_order(c_in, 0, c_out) != 0


def _k3_one_channel(x: torch.Tensor, w: torch.Tensor, s: ConvSpec, **epilogue) -> torch.Tensor:
    """3x3x3 convolution to ONE output channel as in inference (engine.py, 'k3_w'): the dot product of every input row with
    all 27 offset kernels on the MFMA kernel (a pointwise GEMM to 27 -> 32 columns), then 27 gathered scalars per output
    row -- instead of 27 gathered rows per output row on the VALU kernel"""
    c_in = w.shape[-2]
    wt = torch.nn.functional.pad(w.detach().reshape(27, c_in).t(), (0, 5))         # [c_in, 32], columns 27.. zero (one launch)
    y = ops.conv_f32(x, wt, 32, s.n_in, pack=PACK)
    return ops.gather_sum(y, s.table, 27, s.n_in, 1, s.n_in, **epilogue)


def _one_channel_ok(c_in: int, c_out: int) -> bool:
    return c_out == 1 and c_in % 32 == 0


def _forward(x: torch.Tensor, w: torch.Tensor, s: ConvSpec) -> torch.Tensor:
    c_in, c_out = w.shape[-2], w.shape[-1]
    if s.kind == 'k1':
        return ops.conv_f32(x, w.reshape(c_in, c_out), c_out, s.n_in, pack=PACK)
    if s.kind == 'k3' and _one_channel_ok(c_in, c_out):
        return _k3_one_channel(x, w, s)
    if s.kind == 'k3':
        return ops.conv_f32(x, w, c_out, s.n_in, nbr=s.table, n_offsets=27, nbr_ks=s.n_in, nbr_os=1,
                            row_order=s.row_order if _mfma(c_in, c_out) else None, pack=PACK)
    if s.kind == 'k2s2':
        return ops.conv_f32(x, w, c_out, s.n_out, nbr=s.table, n_offsets=8, nbr_ks=1, nbr_os=8)
    if s.kind == 'k2s2T':
        return ops.conv_f32(x, w, c_out, s.n_in, groups=8, out_map=s.table, om_os=8, om_gs=1, out_rows=s.n_out)
    if s.kind == 'gen':
        return ops.conv_f32(x, w, c_out, s.n_in, groups=8)
    if s.kind == 'tab':
        k = s.table.shape[1]
        out = None
        step = ops.table_conv_chunk(c_in, c_out, k)   # the SAME partition as the inference path (int_sparse_conv.Conv3d._run), so
        for a in range(0, k, step):                   # a float model gives the same fp32 bits whether or not grad mode is on
            b = min(a + step, k)
            part = ops.conv_f32(x, w.reshape(k, c_in, c_out)[a:b].contiguous(), c_out, s.n_out,
                                nbr=s.table[:, a:b].contiguous(), n_offsets=b - a, nbr_ks=1, nbr_os=b - a)
            out = part if out is None else out.add_(part)
        return out
    raise ValueError(s.kind)


_TAB_CHUNK = 16       # kernel offsets per weight-gradient launch of the general-table path (a 4x4x4 kernel has 64)


def _wide(call, wt: torch.Tensor, width: int, rows: int, device) -> torch.Tensor:
    """a convolution with more than 128 output columns (the input gradient of a layer that read a 256-channel
    concatenation) as 128-column launches of the MFMA kernel into one output matrix"""
    if width <= 128 or width % 128:
        return call(wt, width, None)
    out = torch.empty((rows, width), dtype=torch.float32, device=device)
    for lo in range(0, width, 128):
        call(wt[..., lo: lo + 128].contiguous(), 128, out[:, lo: lo + 128])
    return out


def _input_grad(dy: torch.Tensor, w: torch.Tensor, s: ConvSpec) -> torch.Tensor:
    c_in, c_out = w.shape[-2], w.shape[-1]
    if s.kind == 'tab':
        raise NotImplementedError('input gradient through a general lookup-table convolution')
    if s.kind == 'k1':
        return _wide(lambda wt, c, out: ops.conv_f32(dy, wt, c, s.n_in, out=out, pack=PACK),
                     ops.transpose_weights(w, 1, c_in, c_out, flip=False).view(c_out, c_in), c_in, s.n_in, dy.device)
    if s.kind == 'k3' and _one_channel_ok(c_in, c_out):
        # Input gradient of a 3x3x3 convolution to ONE channel, the mirror of the forward's two-phase form: the 27 upstream scalars of
        # every row gathered side by side, G[i][k] = dy[nbr[k][i]] (0 where the neighbour is absent), then ONE per-point MFMA GEMM
        # G [n, 32] @ W' [32, c_in] with W'[k] = W[26 - k] -- instead of 27 scalar-times-row products per row on the VALU kernel
        # (k_conv_valu<16>: 1.3 ms per step, profiles/r05/train_host_ops.md).
        # The gather is ONE indexing launch: absent neighbours and the five padding columns point at a zero appended to dy; the index
        # matrix [n, 32] belongs to the map and is built once per map and step (the layers of a level share their table).
        gidx = getattr(s.table, '_fpcc_gather_rows', None)
        if gidx is None:
            t = s.table.t()                                                                           # [n, 27]
            gidx = s.table._fpcc_gather_rows = torch.nn.functional.pad(torch.where(t >= 0, t, s.n_in), (0, 5), value=s.n_in).long()
        g = torch.cat((dy.reshape(-1), dy.new_zeros(1)))[gidx]                                        # [n, 32]
        wt = torch.nn.functional.pad(w.detach().reshape(27, c_in).flip(0), (0, 0, 0, 5))              # [32, c_in], rows 27.. zero
        return ops.conv_f32(g, wt, c_in, s.n_in, pack=PACK)
    if s.kind == 'k3':
        wt = ops.transpose_weights(w, 27, c_in, c_out, flip=True)         # W'[k] = W[26-k]^T
        return _wide(lambda wk, c, out: ops.conv_f32(dy, wk, c, s.n_in, nbr=s.table, n_offsets=27, nbr_ks=s.n_in, nbr_os=1,
                                                     row_order=s.row_order if _mfma(c_out, c) else None, out=out, pack=PACK),
                     wt, c_in, s.n_in, dy.device)
    wt = ops.transpose_weights(w, 8, c_in, c_out, flip=False)               # [8][c_out][c_in]
    if s.kind == 'k2s2':                                                   # children <- parents: transposed form
        return ops.conv_f32(dy, wt, c_in, s.n_out, groups=8, out_map=s.table, om_os=8, om_gs=1, out_rows=s.n_in)
    if s.kind == 'k2s2T':                                                  # parents <- children: strided form
        return ops.conv_f32(dy, wt, c_in, s.n_in, nbr=s.table, n_offsets=8, nbr_ks=1, nbr_os=8)
    if s.kind == 'gen':                                                    # dY [8m, c_out] read as [m, 8*c_out]
        return ops.conv_f32(dy.view(s.n_in, 8 * c_out), wt.reshape(8 * c_out, c_in), c_in, s.n_in)
    raise ValueError(s.kind)


def _weight_grad(x: torch.Tensor, dy: torch.Tensor, w: torch.Tensor, s: ConvSpec) -> torch.Tensor:
    if s.kind == 'k1':
        dw = ops.conv_wgrad(x, dy, s.n_in)
    elif s.kind == 'k3':
        dw = ops.conv_wgrad(x, dy, s.n_in, nbr=s.table, n_offsets=27, nbr_ks=s.n_in, nbr_os=1, row_order=s.row_order)
    elif s.kind == 'k2s2':
        dw = ops.conv_wgrad(x, dy, s.n_out, nbr=s.table, n_offsets=8, nbr_ks=1, nbr_os=8)
    elif s.kind == 'k2s2T':
        dw = ops.conv_wgrad(x, dy, s.n_in, groups=8, out_map=s.table, om_os=8, om_gs=1)
    elif s.kind == 'gen':
        c_in, c_out = w.shape[-2], w.shape[-1]
        if 8 * c_out <= 128 and c_in % 16 == 0 and (8 * c_out) % 32 == 0:
            # a generated set's rows are 8 * parent + octant, so dY [8 m, c_out] IS [m, 8 c_out]: the eight octant gradients side by side
            # are ONE per-point weight gradient X^T [c_in, m] . dY [m, 8 c_out] on the MFMA kernel (the mirror of the packed forward GEMM)
            # instead of eight 16-column ones on the narrow kernels (64 -> 16 at 125 K rows: 0.70 ms at 0.4 TFLOP/s,
            # profiles/r05/train_host_ops.md).  Same sums in the same row order per element.
            wide = ops.conv_wgrad(x, dy.view(s.n_in, 8 * c_out), s.n_in)             # [1, 1, c_in, 8 c_out]
            dw = wide.view(c_in, 8, c_out).permute(1, 0, 2).contiguous()
        else:
            dw = ops.conv_wgrad(x, dy, s.n_in, groups=8)
    elif s.kind == 'tab':
        k = s.table.shape[1]
        dw = torch.cat([ops.conv_wgrad(x, dy, s.n_out, nbr=s.table[:, a: a + _TAB_CHUNK].contiguous(),
                                       n_offsets=min(_TAB_CHUNK, k - a), nbr_ks=1, nbr_os=min(_TAB_CHUNK, k - a))
                        for a in range(0, k, _TAB_CHUNK)], 1)
    else:
        raise ValueError(s.kind)
    return dw.view(w.shape)


class SparseConvFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x: torch.Tensor, w: torch.Tensor, spec: ConvSpec):
        x = x.contiguous()
        w = w.contiguous()
        ctx.save_for_backward(x, w)
        ctx.spec = spec
        return _forward(x, w, spec)

    @staticmethod
    def backward(ctx, dy: torch.Tensor):
        x, w = ctx.saved_tensors
        s = ctx.spec
        dy = dy.contiguous()
        dx = _input_grad(dy, w, s) if ctx.needs_input_grad[0] else None
        dw = _weight_grad(x, dy, w, s) if ctx.needs_input_grad[1] else None
        return dx, dw, None


def sparse_conv(x: torch.Tensor, w: torch.Tensor, spec: ConvSpec) -> torch.Tensor:
    return SparseConvFn.apply(x, w, spec)


class SparseConvActFn(torch.autograd.Function):
    """convolution + bias + (P)ReLU as ONE node: the forward is the fused inference launch, the backward recovers
    dL/d(pre-activation), dbias and dslope from the saved OUTPUT (fpcc_epilogue_bwd_f32; needs a PReLU slope > 0) and
    feeds the two gradient convolutions"""

    @staticmethod
    def forward(ctx, x, w, bias, slope, spec: ConvSpec, act: int):
        x = x.contiguous()
        w = w.contiguous()
        c_in, c_out = w.shape[-2], w.shape[-1]
        b = None if bias is None else bias.reshape(-1)
        kw = dict(bias=b, act=act, slope=slope)
        if spec.kind == 'k1':
            y = ops.conv_f32(x, w.reshape(c_in, c_out), c_out, spec.n_in, pack=PACK, **kw)
        elif spec.kind == 'k3' and _one_channel_ok(c_in, c_out):
            y = _k3_one_channel(x, w, spec, **kw)
        elif spec.kind == 'k3':
            y = ops.conv_f32(x, w, c_out, spec.n_in, nbr=spec.table, n_offsets=27, nbr_ks=spec.n_in, nbr_os=1,
                             row_order=spec.row_order if _mfma(c_in, c_out) else None, pack=PACK, **kw)
        elif spec.kind == 'k2s2':
            y = ops.conv_f32(x, w, c_out, spec.n_out, nbr=spec.table, n_offsets=8, nbr_ks=1, nbr_os=8, **kw)
        elif spec.kind == 'k2s2T':
            y = ops.conv_f32(x, w, c_out, spec.n_in, groups=8, out_map=spec.table, om_os=8, om_gs=1, out_rows=spec.n_out, **kw)
        elif spec.kind == 'gen':
            y = ops.conv_f32(x, w, c_out, spec.n_in, groups=8, **kw)
        else:
            raise ValueError(spec.kind)
        ctx.save_for_backward(x, w, y, slope if slope is not None else x.new_empty(0))
        ctx.spec, ctx.act, ctx.has_bias, ctx.has_slope = spec, act, bias is not None, slope is not None
        ctx.bias_shape = None if bias is None else bias.shape
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y, slope = ctx.saved_tensors
        s = ctx.spec
        want_b = ctx.has_bias and ctx.needs_input_grad[2]
        want_s = ctx.has_slope and ctx.needs_input_grad[3]
        if ctx.act == ops.ACT_NONE and not want_b:
            g, dbias, dslope = dy.contiguous(), None, None
        else:       # the epilogue kernel reads rows at any stride (a column slice of a concatenation's gradient) and writes g packed
            g, dbias, dslope = ops.epilogue_bwd(y, dy if _strided_rows(dy) else dy.contiguous(), ctx.act, slope if ctx.has_slope else None,
                                                want_b, want_s)
        dx = _input_grad(g, w, s) if ctx.needs_input_grad[0] else None
        dw = _weight_grad(x, g, w, s) if ctx.needs_input_grad[1] else None
        if dbias is not None:
            dbias = dbias.view(ctx.bias_shape)
        if dslope is not None:
            dslope = dslope.view(slope.shape)
        return dx, dw, dbias, dslope, None, None


def sparse_conv_act(x, w, bias, slope, spec: ConvSpec, act: int) -> torch.Tensor:
    return SparseConvActFn.apply(x, w, bias, slope, spec, act)


class LinearActFn(torch.autograd.Function):
    """per-point linear layer + bias + (P)ReLU on an nn.Linear weight [c_out, c_in] as it is stored.  Through SparseConvActFn the layer
    cost a transposed weight copy forward, a weight transposition kernel backward (the input gradient wants [c_out, c_in] -- which IS
    the stored layout) and a copy of the weight gradient (computed as [c_in, c_out], handed to autograd through the `.t()` view,
    made contiguous when stored).  Here the input gradient reads the parameter itself and the weight gradient is computed transposed,
    dW [c_out, c_in] = g^T x (the weight-gradient kernel with its operands swapped): two launches and a copy less per layer and step."""

    @staticmethod
    def forward(ctx, x, weight, bias, slope, act: int):
        x = x.contiguous()
        c_out, c_in = weight.shape
        n = x.shape[0]
        y = ops.conv_f32(x, weight.detach().t().contiguous(), c_out, n, bias=None if bias is None else bias.reshape(-1), act=act, slope=slope,
                         pack=PACK)
        ctx.save_for_backward(x, weight, y, slope if slope is not None else x.new_empty(0))
        ctx.act, ctx.has_bias, ctx.has_slope = act, bias is not None, slope is not None
        ctx.bias_shape = None if bias is None else bias.shape
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, y, slope = ctx.saved_tensors
        c_out, c_in = weight.shape
        n = x.shape[0]
        want_b = ctx.has_bias and ctx.needs_input_grad[2]
        want_s = ctx.has_slope and ctx.needs_input_grad[3]
        if ctx.act == ops.ACT_NONE and not want_b:
            g, dbias, dslope = dy.contiguous(), None, None
        else:
            g, dbias, dslope = ops.epilogue_bwd(y, dy if _strided_rows(dy) else dy.contiguous(), ctx.act, slope if ctx.has_slope else None,
                                                want_b, want_s)
        dx = dw = None
        if ctx.needs_input_grad[0]:
            dx = _wide(lambda wt, c, out: ops.conv_f32(g, wt, c, n, out=out, pack=PACK), weight.detach(), c_in, n, dy.device)
        if ctx.needs_input_grad[1]:
            if c_in in (32, 64, 128) and c_out % 64 == 0:        # shapes the matrix-core gradient kernel takes with the operands swapped
                dw = ops.conv_wgrad(g, x, n).view(c_out, c_in)
            else:                                                # (a 256-wide input: its own orientation, transposed as a view)
                dw = ops.conv_wgrad(x, g, n).view(c_in, c_out).t()
        if dbias is not None:
            dbias = dbias.view(ctx.bias_shape)
        if dslope is not None:
            dslope = dslope.view(slope.shape)
        return dx, dw, dbias, dslope, None


def sparse_linear_act(x, weight, bias, slope, act: int) -> torch.Tensor:
    return LinearActFn.apply(x, weight, bias, slope, act)


class BoundFunction(torch.autograd.Function):
    """clamp to [-bound, bound]; the gradient is replaced by +1 / -1 where the value left the interval
    (/root/reference/models/convolutional/lossy_coord_v2/layers.py:13-25)"""

    @staticmethod
    def forward(ctx, x, bound):
        ctx.save_for_backward(x, bound)
        return torch.clip(x, -bound, bound)

    @staticmethod
    def backward(ctx, g):
        x, bound = ctx.saved_tensors
        # +1 right of the interval, -1 left of it, g inside (bound >= 0): four launches, no mask indexing (that would synchronise)
        return torch.where(x.abs() > bound, torch.sign(x), g), None
