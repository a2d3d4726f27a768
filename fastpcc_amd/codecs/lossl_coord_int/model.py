"""Integer-only lossless LiDAR geometry codec (octree occupancy predicted by int8 sparse-conv networks, one rANS stream):
module tree, buffer names and bitstream of /root/reference/models/convolutional/lossl_coord_int/model.py:28-545 on top of
fastpcc_amd.int_sparse_conv.

Data flow kept from the reference: Morton ('zyx') sorted voxels -> 13 octree levels of 8-bit child occupancy (`get_bin`)
-> per level a residual int8 network predicts 255-ary logits (Q8.23) -> LUT softmax -> uint16 CDF -> rANS.
What differs is where bytes move:
  * encode: the CDF rows never leave the GPU; a kernel resolves each coded symbol to (start, freq-1) (4 B per symbol
    over PCIe instead of 510 B) and the host pushes those ranges -- same stream bytes;
  * decode: CDF rows are produced by one fused kernel per level and copied once; the serial rANS chain runs on the host
    as in the reference (it needs the whole row of every symbol);
  * every conv/linear carries its fixed-point epilogue (one launch per layer).
"""
import io
import math
import time
from typing import List, Optional

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import hipops as ops
from ...data import PCData
from ...int_sparse_conv import LinearIn8W8Out8, LinearIn8W8Out32, LinearPReLUIn8W8Out8, LinearPReLUIn8W8Out32, \
    PReLUIn32Out32, RequantFxpToScaledInt8, ROW_ORDER_MIN_ROWS, ROW_ORDER_WINDOW_LOG2, SharedFxpShift, SparseConvIn8W8Out8, \
    SparseConvIn8W8Out32, SparseConvPReLUIn8W8Out8, SparseConvPReLUIn8W8Out32, SparseResBlockIn32W8Out32, SparseTensor, \
    _kernel_table, sparse_conv_in8w8out32
from ...rans_coder import RansDecoder, RansEncoder
from .model_config import Config

_DENSE = (PReLUIn32Out32, RequantFxpToScaledInt8, LinearIn8W8Out8, LinearIn8W8Out32, LinearPReLUIn8W8Out8,
          LinearPReLUIn8W8Out32)
PRE_SHIFT = SharedFxpShift - 16          # Q8.23 logits -> Q15.16 softmax input


class SparseSequential(nn.Sequential):
    """modules of _DENSE act on the feature matrix, the others on the sparse tensor (model.py:524-534)"""

    def forward(self, input: SparseTensor, _also=None) -> SparseTensor:
        """_also: requantisers that will consume the LAST module's output (handed to it when it can write their int8 copies)"""
        x = SparseTensor(input.F, input.C, input.stride, input.spatial_range)
        x._caches = input._caches
        mods = list(self)
        for i, module in enumerate(mods):
            x = _apply(module, x, _consumers(mods, i + 1) or (_also if i == len(mods) - 1 else None))
        return x


def _consumers(mods, at: int):
    """the requantiser that reads the output of mods[at - 1], when the next module is one (or starts with one)"""
    if at >= len(mods):
        return None
    nxt = mods[at]
    if isinstance(nxt, RequantFxpToScaledInt8):
        return [nxt]
    if isinstance(nxt, SparseResBlockIn32W8Out32):
        return [nxt.input_requant]
    return None


def _apply(module, x: SparseTensor, also) -> SparseTensor:
    """one module of a SparseSequential; producers of Q8.23 activations are told their consumers' requantisers"""
    if isinstance(module, _DENSE):
        if also and isinstance(module, (LinearIn8W8Out32, LinearPReLUIn8W8Out32)):
            x.F = module(x.F, _also=also)
        else:
            x.F = module(x.F)
        return x
    if also and isinstance(module, SparseResBlockIn32W8Out32):
        return module(x, _also=also)
    if also and isinstance(module, (SparseConvIn8W8Out32, SparseConvPReLUIn8W8Out32)):
        return module.forward_with_sparse_tensor(x, _also=also)
    return module(x)


class Occupancy:
    """The 8-bit child occupancy of the n voxels of a level (symbols int16 [n]: symbol + 1 = the bits, or a [n, 8] 0/1 matrix) and
    everything the traversal derives from it -- the occupied (row, octant) pairs in row-major order, the gather table of an
    "occupied outputs only" linear layer, the bits as features, the children's coordinates -- produced by ONE kernel behind a
    scan (fpcc_octree_children) instead of the reference's nonzero / index_select / shift / add / cat / scatter operators
    (model.py:60-74,169-175,430-470).  `count` = occupied children in all; known on the host in both directions (encoder: the
    next level's row count, decoder: the popcount of the symbols the host just decoded), so the device is never synchronised."""

    def __init__(self, symbols: Optional[torch.Tensor] = None, bits: Optional[torch.Tensor] = None, count: Optional[int] = None):
        src = symbols if symbols is not None else bits
        self.n = src.shape[0]
        self.symbols = symbols
        self._bits_in = None if bits is None else (bits if bits.dtype == torch.uint8 else (bits != 0).to(torch.uint8)).contiguous()
        if count is None:
            count = getattr(src, '_fpcc_children', None)
        if count is None:                                      # no host-side knowledge: one read-back
            count = int((self._bits_in != 0).sum().item()) if symbols is None else \
                int(((((symbols.to(torch.int32) + 1) & 255)[:, None] >> torch.arange(8, device=src.device)) & 1).sum().item())
        self.count = int(count)
        self._d = None
        self._children = {}

    def _derive(self, coords: Optional[torch.Tensor] = None):
        d = ops.octree_children(self.n, self.count, symbols=self.symbols, bits=self._bits_in,
                                coords=None if coords is None else coords.contiguous(), fxp_one=1 << SharedFxpShift,
                                want_table=self._d is None, want_bits=self._d is None)
        if self._d is None:
            self._d = d
        return d

    def _get(self, key: str) -> torch.Tensor:
        if self._d is None:
            self._derive()
        return self._d[key]

    bits = property(lambda self: self._get('bits'))              # uint8 [n, 8]
    fxp = property(lambda self: self._get('fxp'))                # int32 [n, 8]: the bits as Q8.23 features
    parent_row = property(lambda self: self._get('parent_row'))  # int32 [count]
    octant = property(lambda self: self._get('octant'))          # int32 [count]
    table = property(lambda self: self._get('table'))            # int32 [ceil128(count), 8]

    def bool(self) -> torch.Tensor:
        return self.bits.bool()

    def children(self, coords: torch.Tensor) -> torch.Tensor:
        """[n, 4] coordinates of the level -> [count, 4] coordinates of the occupied children (one level finer)"""
        key = (coords.data_ptr(), coords.shape[0])
        if key not in self._children:
            self._children[key] = self._derive(coords)['child_coords']
        return self._children[key]


def _as_occ(mask, count: Optional[int] = None) -> Occupancy:
    if isinstance(mask, Occupancy):
        return mask
    occ = getattr(mask, '_fpcc_occ', None)                      # a bits matrix seen before (the encoder's levels)
    if occ is None:
        sym = getattr(mask, '_fpcc_symbols', None)
        occ = Occupancy(bits=mask, count=count) if sym is None else Occupancy(symbols=sym, count=count)
        try:
            mask._fpcc_occ = occ
        except AttributeError:
            pass
    return occ


def _occupied_outputs(seq: 'SparseSequential', x: SparseTensor, mask, count: Optional[int] = None, _also=None) -> torch.Tensor:
    """seq(x).F.reshape(n, 8, C)[mask] for a sequence that ends in a linear layer C_in -> 8*C (model.py:66-74,169-175): the
    reference evaluates all 8*C columns of every row and keeps the occupied octants; here the last layer is evaluated for the
    occupied (row, octant) pairs only -- an 8-"offset" gather convolution whose table has one entry per output row, followed by
    the layer's epilogue with the octant's bias / multiplier columns.  Same integers, ~1/6 of the arithmetic and of the
    int32 traffic on a LiDAR sweep."""
    occ = _as_occ(mask, count)
    last = seq[len(seq) - 1] if len(seq) else None
    if not isinstance(last, LinearIn8W8Out32) or last.out_ch % 8:
        f = seq(x).F
        return f.reshape(f.shape[0], 8, f.shape[1] // 8)[occ.parent_row.long(), occ.octant.long()]
    y = SparseTensor(x.F, x.C, x.stride, x.spatial_range)
    y._caches = x._caches
    mods = list(seq)
    for i, module in enumerate(mods[:-1]):
        y = _apply(module, y, _consumers(mods, i + 1))
    n_child, ch = occ.count, last.out_ch // 8
    w = last._padded_weight()                                           # [1, 8*C, ldw] -> [8, C, ldw]
    raw = ops.conv_i8(y.F, w.view(8, ch, w.shape[-1]), last.in_ch, ch, n_child, nbr=occ.table, n_offsets=8, nbr_ks=1, nbr_os=8,
                      nbr_bias=1)
    ep = last._epilogue()
    if _also:                                                        # the next level's first requantiser: written right here
        hints = list(_also)
        out, extra = ops.epilogue_i32(raw, ep['requant_mul'], ep['zero_point'], ep['shift'], ep['out_bits'], bias=ep['bias'],
                                      row_group=occ.octant, also=[h.hint(ch) for h in hints])
        out._fpcc_q8 = {id(h): b for h, b in zip(hints, extra)}
        return out
    return ops.epilogue_i32(raw, ep['requant_mul'], ep['zero_point'], ep['shift'], ep['out_bits'], bias=ep['bias'],
                            row_group=occ.octant)


def _children_of(coords: torch.Tensor, unfold_kernel: torch.Tensor, mask) -> torch.Tensor:
    """[N, 4] level-l coordinates + the level's occupancy -> coordinates of the occupied children at level l-1"""
    return _as_occ(mask).children(coords)


_POPCOUNT8 = np.array([bin(v).count('1') for v in range(256)], dtype=np.int64)


def _children_count(symbols: np.ndarray) -> int:
    """occupied children of the decoded symbols (symbol + 1 = the 8 occupancy bits), counted on the host"""
    if hasattr(np, 'bitwise_count'):                            # numpy >= 2: three narrow passes instead of a 64-bit table look-up
        return int(np.bitwise_count(((symbols.astype(np.uint16) + 1) & 0xff).astype(np.uint8)).sum(dtype=np.int64))
    return int(_POPCOUNT8[(symbols.astype(np.int64) + 1) & 0xff].sum())


def _symbols_of(bits: torch.Tensor, bin2oct: torch.Tensor) -> torch.Tensor:
    """8 occupancy bits (child k = 4dx + 2dy + dz) -> symbol in [0, 254] (model.py:60)"""
    ready = getattr(bits, '_fpcc_symbols', None)                # written beside the bits by the octree analysis (fpcc_octree_level)
    if ready is not None:
        return ready
    return (bits.to(torch.int32) << bin2oct).sum(1, dtype=torch.int32).add_(-1).to(torch.int16)


def _bits_of(symbols: torch.Tensor, bin2oct: torch.Tensor) -> Occupancy:
    return Occupancy(symbols=symbols)          # count: `_fpcc_children`, set where the symbols came from the host decoder


# ---- a level per call -------------------------------------------------------------------------------------------------------------
# The standard integer OneScalePredictor (ten of the thirteen levels of a LiDAR sweep, the seven coarsest of them a few hundred rows
# each) is a fixed sequence of layers: its modules are described ONCE to the library (fpcc_int_onescale) and a level is two calls,
# fpcc_int_level_trunk and fpcc_int_level_expand, instead of ~14 module calls -- the module tree, the state dict and the integers stay
# what they are (every layer still goes through fpcc_conv_i8_also).  FAST_LEVELS = False keeps the layer-by-layer path (tests compare).
FAST_LEVELS = True


def _map27(x: SparseTensor):
    """kernel map of the 3x3x3 / stride-1 convolutions on x's coordinates (and its neighbour-pattern row order on large maps), from or
    into the cloud's caches -- what `_conv_on_sparse_tensor` + `sparse_conv_in8w8out32` do for the first such convolution of a level"""
    caches, tag = x._caches, (x.stride, (3, 3, 3), (1, 1, 1))
    cur = caches.kmaps.get(tag)
    table = cur.get('in_out_maps') if cur is not None else None
    if table is None:
        hashmap, table = _kernel_table(x.C, x.C, (3, 3, 3), (1, 1, 1), caches.hashmaps.get(x.stride))
        caches.kmaps.setdefault(tag, {}).setdefault('in_out_maps', table)
        caches.hashmaps.setdefault(x.stride, hashmap)
        caches.cmaps.setdefault(x.stride, (x.C, x.spatial_range))
    order = getattr(table, '_fpcc_row_order', None)
    if order is None and x.C.shape[0] > ROW_ORDER_MIN_ROWS:
        order = table._fpcc_row_order = ops.conv_row_order(table - 1, 27, 1, 27, x.C.shape[0], ROW_ORDER_WINDOW_LOG2)
    return table, order


class _Described:
    """a ctypes descriptor together with the tensors it points at and the modules it was read from"""

    def __init__(self):
        self.keep, self.layers, self.requants, self.desc = [], [], [], None

    def layer(self, mod):
        w = mod._padded_weight()
        ep = mod._epilogue()
        d = ops.i8_layer(w, mod.in_ch, mod.out_ch, bias=ep['bias'], slope=ep['slope'], requant_mul=ep['requant_mul'],
                         zero_point=ep['zero_point'], shift=ep['shift'], out_bits=ep['out_bits'],
                         zp_comp=mod.int_zero_point_in_comp if getattr(mod, 'use_zero_point_in', False) else None, keep=self.keep)
        self.layers.append((mod, mod._buffers['weight'], mod._buffers['weight']._version, mod._buffers['weight'].data_ptr()))
        return d

    def requant(self, mod):
        mul, zp, shift, _ = mod.hint(0)
        self.requants.append((mod, mod._buffers['requant_mul']))
        return ops.i8_requant(mul, zp, shift, keep=self.keep)

    def __deepcopy__(self, memo):
        return None                # a copied module (serving.clone_context) describes itself again: the table holds raw pointers

    def valid(self) -> bool:
        for mod, w, version, ptr in self.layers:
            if mod._buffers['weight'] is not w or w._version != version or w.data_ptr() != ptr or mod._shift_host is None:
                return False
        for mod, mul in self.requants:
            if mod._buffers['requant_mul'] is not mul or mod._shift_host is None:
                return False
        return True


def _requant_desc(mod: RequantFxpToScaledInt8):
    d = mod.__dict__.get('_fpcc_desc')
    if d is None or not d.valid():
        d = _Described()
        d.desc = d.requant(mod)
        mod.__dict__['_fpcc_desc'] = d
    return d.desc


class OneScalePredictor(nn.Module):
    def __init__(self, channels, if_upsample=True, allow_single_ch=False):
        super().__init__()
        if allow_single_ch:
            self.dec_init = SparseConvIn8W8Out32(1, channels)
        self.dec = SparseResBlockIn32W8Out32(channels)
        self.pred = SparseSequential(RequantFxpToScaledInt8(), SparseConvPReLUIn8W8Out8(channels, channels),
                                     LinearIn8W8Out32(channels, 255))
        self.if_upsample = if_upsample
        self.upsample = SparseSequential(
            RequantFxpToScaledInt8(), LinearPReLUIn8W8Out32(channels + 8, channels), SparseResBlockIn32W8Out32(channels),
            RequantFxpToScaledInt8(), LinearIn8W8Out32(channels, channels * 8)) if if_upsample else None
        self.register_buffer('_shared_fxp_shift', torch.tensor(SharedFxpShift, dtype=torch.int32), persistent=False)

    @staticmethod
    def _feat(bits: torch.Tensor) -> torch.Tensor:
        """occupancy bits as activations: 1.0 in the shared Q8.23 format (the float twin overrides this)"""
        return bits.to(torch.int32) << SharedFxpShift

    def _trunk(self, cur_rec: SparseTensor):
        if cur_rec.F.shape[1] == 1:
            cur_rec = self.dec_init(cur_rec)
        # the block's output R is requantised twice: for `pred` and -- with the occupancy bits appended -- for `upsample`; both
        # int8 copies come out of the block's last epilogue (the second with room for the 8 bit columns, filled in `_expand`)
        also = None
        if isinstance(self.dec, SparseResBlockIn32W8Out32) and len(self.pred) and isinstance(self.pred[0], RequantFxpToScaledInt8):
            also = [self.pred[0]]
            if self.upsample is not None and len(self.upsample) and isinstance(self.upsample[0], RequantFxpToScaledInt8):
                also.append((self.upsample[0], (self.dec.ch + 8 + 15) // 16 * 16))
        cur_rec = self.dec(cur_rec, _also=also) if also else self.dec(cur_rec)
        return cur_rec, self.pred(cur_rec).F

    def _feat_of(self, occ) -> torch.Tensor:
        """occupancy as input features: straight from the octree kernel in the shared fixed-point format, through `_feat` where a
        subclass encodes them differently (the float twin)"""
        occ = _as_occ(occ)
        return occ.fxp if self._feat is OneScalePredictor._feat else self._feat(occ.bits)

    def _expand(self, cur_rec: SparseTensor, bits, child_coords: torch.Tensor, _also=None) -> SparseTensor:
        occ = _as_occ(bits, child_coords.shape[0])
        ready = getattr(cur_rec.F, '_fpcc_q8', None)
        first = self.upsample[0] if len(self.upsample) else None
        if ready is not None and id(first) in ready and self._feat is OneScalePredictor._feat:
            # requant(cat(R, bits << 23)) = [requant(R) | requant(bits << 23)]: the left part was written by the trunk's last
            # epilogue, the eight bit columns are filled here; the int32 concatenation is never built
            q = ready[id(first)]
            mul, zp, shift, _ = first.hint(q.shape[1])
            ops.fill_bits_i8(occ.bits, q, cur_rec.F.shape[1], mul, zp, shift, 1 << SharedFxpShift)
        else:
            cur_rec.F = torch.cat((cur_rec.F, self._feat_of(occ)), 1)
        feats = _occupied_outputs(self.upsample, cur_rec, occ, _also=_also)
        return SparseTensor(feats, child_coords, tuple(s // 2 for s in cur_rec.stride))

    @staticmethod
    def _first_requant_of(block):
        """the requantiser that will read this level's output features in `block` (the next finer level's predictor), if it reads
        them as they are (a one-scale predictor's residual block); None otherwise"""
        dec = getattr(block, 'dec', None)
        if isinstance(block, OneScalePredictor) and isinstance(dec, SparseResBlockIn32W8Out32):
            return [dec.input_requant]
        return None

    # -- the level-per-call path ---------------------------------------------------------------------------------------------------
    def _described(self):
        """this block as a fpcc_int_onescale, or None when it is not the standard integer block (the float twin, other layer types,
        a width that is not a multiple of 16): then the layer-by-layer path runs"""
        if not FAST_LEVELS or self._feat is not OneScalePredictor._feat:
            return None
        d = self.__dict__.get('_fpcc_desc')
        if d is not None and (d is False or d.valid()):
            return d or None
        self.__dict__['_fpcc_desc'] = d = self._describe() or False
        return d or None

    def _describe(self):
        dec, pred, up = self.dec, self.pred, self.upsample
        c = getattr(dec, 'ch', 0)
        conv3 = lambda m, cls: type(m) is cls and m.kernel_size == (3, 3, 3) and m.stride == (1, 1, 1) and (m.in_ch, m.out_ch) == (c, c)
        block = lambda b: type(b) is SparseResBlockIn32W8Out32 and b.ch == c and conv3(b.conv_prelu, SparseConvPReLUIn8W8Out8) and \
            conv3(b.conv2, SparseConvIn8W8Out32)
        if c < 16 or c % 16 or not block(dec) or len(pred) != 3 or type(pred[0]) is not RequantFxpToScaledInt8 or \
                not conv3(pred[1], SparseConvPReLUIn8W8Out8) or type(pred[2]) is not LinearIn8W8Out32 or pred[2].in_ch != c:
            return None
        if up is not None and (len(up) != 5 or type(up[0]) is not RequantFxpToScaledInt8 or type(up[1]) is not LinearPReLUIn8W8Out32 or
                               (up[1].in_ch, up[1].out_ch) != (c + 8, c) or not block(up[2]) or type(up[3]) is not RequantFxpToScaledInt8 or
                               type(up[4]) is not LinearIn8W8Out32 or (up[4].in_ch, up[4].out_ch) != (c, 8 * c)):
            return None
        d = _Described()
        t = ops.IntOneScale()
        t.channels, t.has_upsample = c, int(up is not None)
        t.dec_in, t.dec_conv1, t.dec_conv2 = d.requant(dec.input_requant), d.layer(dec.conv_prelu), d.layer(dec.conv2)
        t.dec_slope = dec.prelu.slope.data_ptr()
        t.pred_in, t.pred_conv, t.pred_linear = d.requant(pred[0]), d.layer(pred[1]), d.layer(pred[2])
        d.keep.append(dec.prelu.slope)
        if up is not None:
            t.up_in, t.up_linear = d.requant(up[0]), d.layer(up[1])
            t.up_res_in, t.up_conv1, t.up_conv2 = d.requant(up[2].input_requant), d.layer(up[2].conv_prelu), d.layer(up[2].conv2)
            t.up_slope = up[2].prelu.slope.data_ptr()
            t.up_out_in, t.up_out = d.requant(up[3]), d.layer(up[4])
            d.keep.append(up[2].prelu.slope)
        d.slopes = [(dec.prelu, dec.prelu._buffers['slope'])] + ([(up[2].prelu, up[2].prelu._buffers['slope'])] if up is not None else [])
        base_valid = d.valid
        d.valid = lambda: base_valid() and all(m._buffers['slope'] is s for m, s in d.slopes)
        d.desc = t
        return d

    def _trunk_fast(self, d, cur_rec: SparseTensor):
        if cur_rec.F.shape[1] == 1:
            cur_rec = self.dec_init(cur_rec)
        feat = cur_rec.F if cur_rec.F.is_contiguous() else cur_rec.F.contiguous()
        ready = getattr(cur_rec.F, '_fpcc_q8', None)
        table, order = _map27(cur_rec)
        res, q_pred, q_up, logits = ops.int_level_trunk(d.desc, feat.shape[0], feat, None if ready is None else ready.get(id(self.dec.input_requant)),
                                                        table, order, self.upsample is not None)
        res._fpcc_q8 = {id(self.pred[0]): q_pred} if q_up is None else {id(self.pred[0]): q_pred, id(self.upsample[0]): q_up}
        out = SparseTensor(res, cur_rec.C, cur_rec.stride, cur_rec.spatial_range)
        out._caches = cur_rec._caches
        return out, logits

    def _expand_fast(self, d, cur_rec: SparseTensor, symbols: torch.Tensor, count: int, coords: Optional[torch.Tensor],
                     child_coords: Optional[torch.Tensor], next_block) -> SparseTensor:
        """coords: the level's coordinates when the children's are to be derived (decoder), else child_coords are given (encoder)"""
        nxt = self._first_requant_of(next_block)
        table, order = _map27(cur_rec)
        feat, feat_q8, child = ops.int_level_expand(d.desc, cur_rec.F.shape[0], count, cur_rec.F, cur_rec.F._fpcc_q8[id(self.upsample[0])],
                                                    symbols, coords, table, order, None if not nxt else _requant_desc(nxt[0]))
        if feat_q8 is not None:
            feat._fpcc_q8 = {id(nxt[0]): feat_q8}
        return SparseTensor(feat, child if coords is not None else child_coords, tuple(s // 2 for s in cur_rec.stride))

    def compress(self, cur_rec, up_ref: SparseTensor, cur_bin, bin2oct_kernel, if_upsample, next_block=None):
        d = self._described()
        if d is not None:
            cur_rec, cur_pred = self._trunk_fast(d, cur_rec)
            cur_oct = _symbols_of(cur_bin, bin2oct_kernel).contiguous()
            if if_upsample:
                cur_rec = self._expand_fast(d, cur_rec, cur_oct, up_ref.C.shape[0], None, up_ref.C, next_block)
                cur_rec._caches = up_ref._caches
            return cur_rec, cur_pred, cur_oct
        cur_rec, cur_pred = self._trunk(cur_rec)
        cur_oct = _symbols_of(cur_bin, bin2oct_kernel)
        if if_upsample:
            cur_rec = self._expand(cur_rec, Occupancy(symbols=cur_oct, count=up_ref.C.shape[0]), up_ref.C,
                                   _also=self._first_requant_of(next_block))
            cur_rec._caches = up_ref._caches
        return cur_rec, cur_pred, cur_oct

    def decompress(self, cur_rec, bin2oct_kernel, unfold_kernel, rans_decode_oct, if_upsample, next_block=None):
        d = self._described()
        if d is not None:
            cur_rec, cur_pred = self._trunk_fast(d, cur_rec)
            symbols = rans_decode_oct(cur_pred)
            cur_bin = _bits_of(symbols, bin2oct_kernel)
            if if_upsample:         # (a fresh cache namespace per decoded level, as the layer-by-layer path has it)
                cur_rec = self._expand_fast(d, cur_rec, symbols.contiguous(), cur_bin.count, cur_rec.C.contiguous(), None, next_block)
            return cur_rec, cur_bin
        cur_rec, cur_pred = self._trunk(cur_rec)
        cur_bin = _bits_of(rans_decode_oct(cur_pred), bin2oct_kernel)
        if if_upsample:
            cur_rec = self._expand(cur_rec, cur_bin, _children_of(cur_rec.C, unfold_kernel, cur_bin),
                                   _also=self._first_requant_of(next_block))
        return cur_rec, cur_bin


class OneScaleMultiStepPredictor(nn.Module):
    def __init__(self, channels, pred_steps=2, use_more_ch_for_multi_step_pred=True):
        super().__init__()
        self.pred_steps = pred_steps
        span = (2 ** (pred_steps - 2),) * 3
        if pred_steps == 2:
            self.embed = SparseSequential()
            out_ch = channels
            self.dec = SparseSequential(RequantFxpToScaledInt8(), LinearPReLUIn8W8Out32(channels + 8, out_ch),
                                        SparseResBlockIn32W8Out32(out_ch))
        elif use_more_ch_for_multi_step_pred:
            if pred_steps == 3:
                emb, in_ch, out_ch = 64, channels + 64, round(channels * 1.25)
            elif pred_steps >= 4:
                emb, in_ch, out_ch = 512, round(channels * 1.25) + 512, channels * 2
            else:
                raise NotImplementedError
            self.embed = SparseSequential(RequantFxpToScaledInt8(), SparseConvPReLUIn8W8Out32(8, emb, span, span))
            self.dec = SparseSequential(RequantFxpToScaledInt8(), LinearPReLUIn8W8Out32(in_ch, out_ch),
                                        SparseResBlockIn32W8Out32(out_ch)) if in_ch != out_ch else \
                SparseResBlockIn32W8Out32(out_ch)
        else:
            if pred_steps < 3:
                raise ValueError(pred_steps)
            conv = SparseConvPReLUIn8W8Out32 if channels >= 256 else SparseConvIn8W8Out32
            self.embed = SparseSequential(RequantFxpToScaledInt8(), conv(8, channels, span, span))
            self.dec = SparseSequential(RequantFxpToScaledInt8(), LinearPReLUIn8W8Out32(channels + channels, channels),
                                        SparseResBlockIn32W8Out32(channels))
            out_ch = channels
        self.pred = nn.ModuleList()
        for i in range(pred_steps):
            if i == 0:
                self.pred.append(SparseSequential(RequantFxpToScaledInt8(), SparseConvPReLUIn8W8Out8(out_ch, out_ch),
                                                  LinearIn8W8Out32(out_ch, channels * 8)))
            elif i != pred_steps - 1:
                self.pred.append(SparseSequential(PReLUIn32Out32(), RequantFxpToScaledInt8(),
                                                  LinearPReLUIn8W8Out8(channels + 8, channels),
                                                  SparseConvPReLUIn8W8Out8(channels, channels),
                                                  LinearIn8W8Out32(channels, channels * 8)))
            else:
                self.pred.append(SparseSequential(RequantFxpToScaledInt8(), SparseConvPReLUIn8W8Out8(channels, channels),
                                                  LinearIn8W8Out32(channels, 255)))
        self.register_buffer('_shared_fxp_shift', torch.tensor(SharedFxpShift, dtype=torch.int32), persistent=False)

    _feat = staticmethod(OneScalePredictor._feat)
    _feat_of = OneScalePredictor._feat_of

    def _refresh(self, cur_rec: SparseTensor, embed_in: SparseTensor) -> SparseTensor:
        embed_in._caches = cur_rec._caches
        cur_rec.F = torch.cat([cur_rec.F, self.embed(embed_in).F], 1)
        first = self.pred[0][0] if len(self.pred) and len(self.pred[0]) else None
        if isinstance(first, RequantFxpToScaledInt8) and isinstance(self.dec, (SparseSequential, SparseResBlockIn32W8Out32)):
            return self.dec(cur_rec, _also=[first])                  # pred[0]'s requantiser: written by the block's last epilogue
        return self.dec(cur_rec)

    def _descend(self, cur_rec: SparseTensor, masks: List[torch.Tensor], bits_below: List[torch.Tensor],
                 coords: List[torch.Tensor], strides: List[tuple]) -> torch.Tensor:
        """pred[0] on the feature level, then one refinement per finer level: masks[i] selects the occupied children of
        step i, bits_below[i] (absent for the last step) are the occupancy bits appended as extra channels."""
        cur, last = cur_rec, len(self.pred) - 1
        for i in range(1, last + 1):
            f = _occupied_outputs(self.pred[i - 1], cur, masks[i - 1], coords[i - 1].shape[0])
            if i != last:
                f = torch.cat([f, self._feat_of(bits_below[i - 1])], 1)
            cur = SparseTensor(f, coords[i - 1], strides[i - 1])
            cur._caches = cur_rec._caches
        return self.pred[last](cur).F

    def compress(self, cur_rec: SparseTensor, cur_bins: List[SparseTensor], bin2oct_kernel):
        # occupancy of cur_bins[j] (j >= 1): its children are the voxels of cur_bins[j - 1]
        occs = [None] + [_as_occ(cur_bins[j].F, cur_bins[j - 1].C.shape[0]) for j in range(1, len(cur_bins))]
        embed_in = SparseTensor(self._feat_of(occs[1]), cur_bins[1].C, stride=cur_bins[1].stride)
        cur_rec = self._refresh(cur_rec, embed_in)
        n = len(self.pred)
        masks = [occs[len(cur_bins) - i] for i in range(1, n)]
        below = [occs[len(cur_bins) - i - 1] if len(cur_bins) - i - 1 >= 1 else None for i in range(1, n)]
        coords = [cur_bins[-i - 1].C for i in range(1, n)]
        strides = [cur_bins[-i - 1].stride for i in range(1, n)]
        logits = self._descend(cur_rec, masks, below, coords, strides)
        return cur_rec, logits, _symbols_of(cur_bins[0].F, bin2oct_kernel)

    def decompress(self, cur_rec: SparseTensor, cur_bins: List[torch.Tensor], top_rec, top_stride, bin2oct_kernel,
                   unfold_kernel, rans_decode_oct):
        if len(cur_bins) == 1:
            top_rec, top_stride = cur_rec.C, cur_rec.stride[0]
        top_rec = _children_of(top_rec, unfold_kernel, cur_bins[-1])
        top_stride //= 2
        caches = cur_rec._caches
        caches.cmaps[(top_stride,) * 3] = (top_rec, None)
        embed_in = SparseTensor(self._feat_of(cur_bins[-1]), caches.cmaps[(top_stride * 2,) * 3][0],
                                stride=(top_stride * 2,) * 3)
        cur_rec = self._refresh(cur_rec, embed_in)
        n = len(self.pred)
        strides, s = [], cur_rec.stride
        for _ in range(1, n):
            s = tuple(v // 2 for v in s)
            strides.append(s)
        coords = [caches.cmaps[st][0] for st in strides]
        masks = [cur_bins[i - 1] for i in range(1, n)]
        below = [cur_bins[i] if i < len(cur_bins) else None for i in range(1, n)]
        logits = self._descend(cur_rec, masks, below, coords, strides)
        cur_bin = _bits_of(rans_decode_oct(logits), bin2oct_kernel)
        return cur_rec, cur_bin, top_rec, top_stride


_HOST_THREADS = None
_HOST_THREADS_LOCK = __import__('threading').Lock()


class Model(nn.Module):
    one_scale_cls, multi_step_cls = OneScalePredictor, OneScaleMultiStepPredictor

    def __init__(self, cfg: Config, device='cuda'):
        super().__init__()
        self.cfg, self.device = cfg, device
        self.max_downsample_times_wo_recurrent = int(np.log2(cfg.max_stride_wo_recurrent))
        self.max_downsample_times = int(np.log2(cfg.max_stride))
        if cfg.fea_stride < 2:
            raise ValueError('fea_stride must be at least 2')
        self.blocks_dec = nn.ModuleList()
        for idx in range(self.max_downsample_times_wo_recurrent):
            steps = int(np.log2(cfg.fea_stride)) - idx
            if steps < 1:
                self.blocks_dec.append(self.one_scale_cls(cfg.channels, True, False))
            elif steps == 1:
                self.blocks_dec.append(self.one_scale_cls(cfg.channels, False, False))
            else:
                self.blocks_dec.append(self.multi_step_cls(cfg.channels, steps, cfg.use_more_ch_for_multi_step_pred))
        self.block_dec_recurrent = self.one_scale_cls(cfg.channels, True, True)
        fold = torch.zeros(8, 8, 1, dtype=torch.int8)
        fold.reshape(8, 8)[...] = torch.eye(8, dtype=torch.int8)
        self.register_buffer('fold2bin_kernel', fold, persistent=False)
        self.register_buffer('bin2oct_kernel', torch.arange(7, -1, -1, dtype=torch.int32), persistent=False)
        self.register_buffer('unfold_kernel', torch.tensor(
            [(0, dx, dy, dz) for dx in (0, 1) for dy in (0, 1) for dz in (0, 1)], dtype=torch.int32)[None], persistent=False)
        # fixed near-uniform CDFs of the side information (model.py:254-257)
        self.fea_side_info_cdf1 = np.arange(2, 65537, dtype=np.int64).astype(np.uint16)[None].copy()
        self.fea_side_info_cdf2 = (np.arange(1, 129, dtype=np.int64) * 512).astype(np.uint16)[None].copy()
        self.fea_side_info_cdf1[:, -1] = 65535
        self.fea_side_info_cdf2[:, -1] = 65535
        self.rans_encoder = RansEncoder(32 * 1024 * 1024)
        self.rans_decoder = RansDecoder()
        # True: the occupancy symbols of every level are decoded by fpcc_simple_dec_pop_dev (CDF rows never cross PCIe);
        # slower than the chunked host path on this codec (profiles/r02/device_rans.md), hence off by default
        self.device_decoder = False
        self._clouds = None            # decompress_many: {'decoders', 'rows' (per cloud, of the level being decoded), 'threads'}
        self._extra_encoders: List[RansEncoder] = []

    # ---------------------------------------------------------------------------------------------------------------
    def forward(self, pc_data: PCData):
        if self.training:
            raise NotImplementedError
        if pc_data.batch_size != 1:
            raise ValueError('Only supports batch size == 1 during testing.')
        return self.test_forward(pc_data)

    @staticmethod
    def get_init_pc(xyz: torch.Tensor, stride: int = 1) -> SparseTensor:
        # coordinates are Morton ('zyx') sorted and unique
        return SparseTensor(torch.ones((xyz.shape[0], 1), dtype=torch.int8, device=xyz.device), xyz, (stride,) * 3)

    @torch.no_grad()
    def get_bin(self, input: SparseTensor, ones_feats: torch.Tensor) -> SparseTensor:
        """next octree level: parent coordinates and the 8 child-occupancy bits per parent (model.py:262-295)"""
        tag = (input.stride, (2, 2, 2), (2, 2, 2))
        out_coords = input.C.clone()
        out_coords[:, 1:] >>= 1
        out_coords = torch.unique_consecutive(out_coords, dim=0)
        out_stride = tuple(s * 2 for s in input.stride)
        bits, hashmap_kv, in_out_maps = sparse_conv_in8w8out32(
            ones_feats[:input.C.shape[0]], self.fold2bin_kernel, input.C, out_coords, (2, 2, 2), (2, 2, 2),
            if_in_coords_equals_out_coords=True)
        caches = input._caches
        if input.stride != (1, 1, 1):
            caches.kmaps.setdefault(tag, {}).setdefault('in_out_maps', in_out_maps)
            caches.hashmaps.setdefault(input.stride, hashmap_kv)
            caches.cmaps.setdefault(input.stride, (input.C, input.spatial_range))
        caches.cmaps.setdefault(out_stride, (out_coords, None))
        ret = SparseTensor(bits, out_coords, out_stride, None)
        ret._caches = caches
        return ret

    @torch.no_grad()
    def analyse(self, xyz: torch.Tensor, keys: torch.Tensor, levels: int, clouds: int = 1):
        """The encoder's octree of a Morton-sorted cloud: what `levels` calls of `get_bin` return (the coordinate levels with their
        child-occupancy bits, the (2, 2, 2) / stride-2 kernel maps and coordinate maps in the cloud's caches), from the sorted keys
        instead of coordinate rows: one counting pass tells the rows of every level (the traversal's only read-back, where
        unique_consecutive synchronised once per level), then each level is one call -- head flags, a scan and one kernel that
        writes coordinates, bits, kernel map and coded symbols (fpcc_octree_level) -- against ~21 operators.
        keys: `cloud << 48 | Morton('zyx')` of xyz's rows, ascending and -- like the voxels of the reference's data sets -- unique (a
        repeated voxel is reported by the counting pass as a ValueError).  -> (levels of SparseTensor, rows[level][cloud])"""
        org = self.get_init_pc(xyz, 1)
        rows = ops.level_counts_clouds(keys, levels, 48, clouds)
        caches = org._caches
        strided, cur = [org], keys
        for l in range(1, levels + 1):
            d = ops.octree_level(cur, sum(rows[l]), 48 - 3 * l)
            prev, stride = strided[-1], (2 ** l,) * 3
            if prev.stride != (1, 1, 1):
                caches.kmaps.setdefault((prev.stride, (2, 2, 2), (2, 2, 2)), {}).setdefault('in_out_maps', d['table'])
                caches.cmaps.setdefault(prev.stride, (prev.C, prev.spatial_range))
            caches.cmaps.setdefault(stride, (d['coords'], None))
            bits = d['bits']
            bits._fpcc_symbols = d['symbols']
            level = SparseTensor(bits, d['coords'], stride, None)
            level._caches = caches
            strided.append(level)
            cur = d['keys']
        return strided, rows

    # -- entropy coding ------------------------------------------------------------------------------------------------
    DECODE_CHUNK_ROWS = 16384      # 8 MB of CDF rows per device->host copy

    def rans_decode_oct(self, logits: torch.Tensor) -> torch.Tensor:
        """CDF rows of the level (510 B per symbol) -> host -> serial rANS decode -> symbols back on the device.  The rows
        cross PCIe in chunks and the decoder works on chunk i while chunk i+1 is in flight: the copy (3.7 ms per frame at
        ~55 GB/s) hides behind the decode instead of preceding it."""
        rows_d = ops.logits_to_cdf16(logits.contiguous(), PRE_SHIFT)
        n = rows_d.shape[0]
        if self.device_decoder:
            # the CDF rows stay where they were made: one wave decodes the level's symbols on the device, 4 bytes come back
            sym, children = ops.simple_dec_pop_dev(self._dev_state, self._dev_stream, self._dev_stream_len, rows_d)
            count, status = children.tolist()                 # the level's one read-back: children + the decoder's status word
            if status != 0:
                raise ValueError(f'corrupt occupancy stream (device decoder status {status})')
            sym._fpcc_children = int(count)
            return sym
        rows_h = torch.empty(rows_d.shape, dtype=rows_d.dtype, pin_memory=True)
        out_h = torch.empty(n, dtype=torch.int16, pin_memory=True)
        # chunk sizes grow 2 Ki, 4 Ki, 8 Ki, then DECODE_CHUNK_ROWS: the host starts decoding after 1 MB has crossed instead of 8 MB
        # (the first chunk is pure latency on the level's critical path), the later chunks keep the per-copy overhead small
        edges, step = [0], min(2048, self.DECODE_CHUNK_ROWS)
        while edges[-1] < n:
            edges.append(min(n, edges[-1] + step))
            step = min(2 * step, self.DECODE_CHUNK_ROWS)
        events = []
        for a, b in zip(edges[:-1], edges[1:]):
            rows_h[a:b].copy_(rows_d[a:b], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            events.append(ev)
        rows_np, out_np = rows_h.numpy().view(np.uint16), out_h.numpy().view(np.uint16)
        clouds = self._clouds
        if clouds is None:
            for (a, b), ev in zip(zip(edges[:-1], edges[1:]), events):
                ev.synchronize()
                self.rans_decoder.decode(rows_np[a:b], out_np[a:b])
        else:
            # several clouds in one traversal (decompress_many): the level's rows are cloud-major, cloud c's symbols come from its own
            # stream -- one host thread per cloud, each waiting only for the copies that cover its rows
            if n != sum(clouds['rows']):
                raise RuntimeError('the level does not hold the rows the clouds\' streams account for')
            starts = np.concatenate(([0], np.cumsum(clouds['rows'])))

            def decode_cloud(c: int) -> int:
                lo, hi = int(starts[c]), int(starts[c + 1])
                for (a, b), ev in zip(zip(edges[:-1], edges[1:]), events):
                    a2, b2 = max(a, lo), min(b, hi)
                    if a2 < b2:
                        ev.synchronize()
                        clouds['decoders'][c].decode(rows_np[a2:b2], out_np[a2:b2])
                return _children_count(out_h.numpy()[lo:hi])
            clouds['rows'] = list(clouds['threads'].map(decode_cloud, range(len(clouds['rows']))))
        out = out_h.to(logits.device, non_blocking=True)
        out._fpcc_children = _children_count(out_h.numpy()) if clouds is None else sum(clouds['rows'])
        return out

    def rans_encode_fea(self, quantized_cdf: np.ndarray, rounded: np.ndarray, encoder: Optional[RansEncoder] = None):
        enc = encoder or self.rans_encoder
        enc.encode(quantized_cdf[None], rounded)
        enc.encode(self.fea_side_info_cdf1, quantized_cdf[:-1] - 1)
        if len(quantized_cdf) - 2 > self.fea_side_info_cdf2.shape[1]:
            raise ValueError('bottom coordinate alphabet too large')
        enc.encode(self.fea_side_info_cdf2, np.array((len(quantized_cdf) - 2,), dtype=np.uint16))

    def rans_decode_fea(self, length: int, decoder: Optional[RansDecoder] = None) -> np.ndarray:
        dec = decoder or self.rans_decoder
        cdf_len = np.empty(1, dtype=np.uint16)
        dec.decode(self.fea_side_info_cdf2, cdf_len)
        cdf = np.empty(int(cdf_len[0]) + 1, dtype=np.uint16)
        dec.decode(self.fea_side_info_cdf1, cdf)
        cdf = np.pad(cdf + 1, (0, 1))
        cdf[-1] = 65535
        decoded = np.empty(length, dtype=np.uint16)
        dec.decode(cdf[None], decoded)
        return decoded

    @staticmethod
    def bottom_cdf(values: np.ndarray) -> np.ndarray:
        """integer-only CDF of the coarsest level's coordinates (model.py:407-415)"""
        counts = np.bincount(values.astype(np.int64), minlength=2).astype(np.int64)
        f = ((counts * (((65536 - counts.shape[0]) << 8) // values.size)) >> 8) + 1
        cdf = np.cumsum(f)
        cdf[-1] = 65535
        return cdf.astype(np.uint16)

    # -- codec ---------------------------------------------------------------------------------------------------------
    def _block(self, idx: int, blocks):
        return self.block_dec_recurrent if idx > len(blocks) else blocks[idx - 1]

    @ops.no_gc_pause
    @torch.no_grad()
    def compress(self, xyz: torch.Tensor) -> bytes:
        if not xyz.is_cuda:
            raise RuntimeError('compress() runs on the GPU; move the coordinates there first')
        coord_offset = xyz.amin(0)[1:]
        xyz = xyz - F.pad(coord_offset, (1, 0))
        keys, perm = ops.sort_keys(ops.morton3d_encode(xyz[:, 1:], (2, 1, 0)))    # 'zyx': z on Morton bit 0
        xyz = xyz[perm.long()].contiguous()
        skip = self.cfg.skip_top_scales_num
        blocks = self.blocks_dec[skip:]
        levels = self.max_downsample_times - skip
        strided, _ = self.analyse(xyz, keys, levels)
        org, top = strided[0], strided[-1]
        bottom_vals = top.C[:, 1:].reshape(-1)
        cur_rec = SparseTensor(org.F[:top.C.shape[0]], top.C, (2 ** levels,) * 3)
        cur_rec._caches = org._caches

        pending = []          # per level, coarse to fine: device (start, freq-1) of its symbols
        for idx in range(levels, 0, -1):
            block = self._block(idx, blocks)
            if isinstance(block, OneScalePredictor):
                cur_rec, logits, symbols = block.compress(cur_rec, strided[idx - 1], strided[idx].F, self.bin2oct_kernel,
                                                          if_upsample=idx != 1 and block.if_upsample,
                                                          next_block=self._block(idx - 1, blocks) if idx > 1 else None)
            else:
                cur_rec, logits, symbols = block.compress(cur_rec, strided[idx: idx + block.pred_steps], self.bin2oct_kernel)
            pending.append(ops.logits_to_ranges(logits.contiguous(), PRE_SHIFT, symbols.contiguous()))

        # one synchronising transfer: 4 bytes per symbol + the coarsest coordinates
        sizes = [s.shape[0] for s, _ in pending]
        start_h = torch.empty(sum(sizes), dtype=torch.int16, pin_memory=True)
        freq_h = torch.empty(sum(sizes), dtype=torch.int16, pin_memory=True)
        start_h.copy_(torch.cat([s for s, _ in pending]), non_blocking=True)
        freq_h.copy_(torch.cat([f for _, f in pending]), non_blocking=True)
        bottom_h = torch.empty(bottom_vals.shape, dtype=torch.int32, pin_memory=True)
        bottom_h.copy_(bottom_vals, non_blocking=True)
        offset_h = coord_offset.cpu()
        torch.cuda.current_stream().synchronize()

        start_np, freq_np = start_h.numpy().view(np.uint16), freq_h.numpy().view(np.uint16)
        edges = np.concatenate(([0], np.cumsum(sizes)))
        for lvl in range(len(sizes) - 1, -1, -1):                 # finest level first: the decoder pops coarse -> fine
            a, b = edges[lvl], edges[lvl + 1]
            self.rans_encoder.encode_ranges(start_np[a:b], freq_np[a:b])
        bottom_np = bottom_h.numpy()
        self.rans_encode_fea(self.bottom_cdf(bottom_np), bottom_np.astype(np.uint16))

        with io.BytesIO() as bs:
            for v in offset_h.tolist():
                bs.write(int(v).to_bytes(2, 'little'))
            bs.write((bottom_np.shape[0] // 3).to_bytes(2, 'little'))
            bs.write(self.rans_encoder.flush())
            return bs.getvalue()

    # -- several clouds in one traversal -------------------------------------------------------------------------------------------
    MANY_MAX_VOXELS = 2_000_000

    def _groups(self, sizes: List[int]) -> List[List[int]]:
        groups, cur, acc = [], [], 0
        for i, n in enumerate(sizes):
            if cur and (acc + n > self.MANY_MAX_VOXELS or len(cur) == 64):
                groups.append(cur)
                cur, acc = [], 0
            cur.append(i)
            acc += n
        if cur:
            groups.append(cur)
        return groups

    def _thread_pool(self, n: int):
        """host threads that code the clouds of a batch side by side (one pool per process: the jobs are short and independent)"""
        global _HOST_THREADS
        from concurrent.futures import ThreadPoolExecutor
        with _HOST_THREADS_LOCK:
            if _HOST_THREADS is None or _HOST_THREADS._max_workers < n:
                _HOST_THREADS = ThreadPoolExecutor(max_workers=max(n, 8), thread_name_prefix='fpcc-int-coder')
            return _HOST_THREADS

    @ops.no_gc_pause
    @torch.no_grad()
    def compress_many(self, clouds: List[torch.Tensor]) -> List[bytes]:
        """B independent clouds (each int32 [n, 4], batch column 0) -> the stream `compress` writes for each, from ONE traversal of the
        octree networks: the clouds become the samples of one batch (the hash-table kernel maps key on the batch column, so rows of
        different clouds never neighbour; integer sums are exact in any order), every operator is launched once per level over all
        clouds, and each cloud's (start, freq) pairs go to its own rANS stream, the clouds' streams coded side by side on host
        threads.  The codec's launches are small (a LiDAR sweep has 113 K voxels on its finest level and 13 levels of ~60
        operators each): this is what fills them.  What the reference does with a list one cloud at a time
        (/root/reference/models/convolutional/lossl_coord_int/model.py compress_partitions)."""
        B = len(clouds)
        if B == 1:
            return [self.compress(clouds[0])]
        if B == 0 or not all(c.is_cuda for c in clouds):
            raise RuntimeError('compress_many() takes a non-empty list of GPU tensors')
        offsets = torch.stack([c.amin(0) for c in clouds])                  # [B, 4]; batch column 0
        shift = offsets.clone()
        shift[:, 0] = -torch.arange(B, device=shift.device, dtype=shift.dtype)
        xyz = torch.cat([c - shift[b] for b, c in enumerate(clouds)])       # cloud b = sample b
        keys = ops.morton3d_encode(xyz[:, 1:], (2, 1, 0)) | (xyz[:, 0].to(torch.int64) << 48)      # 'zyx' Morton inside a cloud
        keys, perm = ops.sort_keys(keys)
        xyz = xyz[perm.long()].contiguous()
        skip = self.cfg.skip_top_scales_num
        blocks = self.blocks_dec[skip:]
        levels = self.max_downsample_times - skip
        strided, level_rows = self.analyse(xyz, keys, levels, B)           # level_rows[l][c]: rows of cloud c on level l
        org, top = strided[0], strided[-1]
        cur_rec = SparseTensor(org.F[:top.C.shape[0]], top.C, (2 ** levels,) * 3)
        cur_rec._caches = org._caches
        pending = []
        for idx in range(levels, 0, -1):
            block = self._block(idx, blocks)
            if isinstance(block, OneScalePredictor):
                cur_rec, logits, symbols = block.compress(cur_rec, strided[idx - 1], strided[idx].F, self.bin2oct_kernel,
                                                          if_upsample=idx != 1 and block.if_upsample,
                                                          next_block=self._block(idx - 1, blocks) if idx > 1 else None)
            else:
                cur_rec, logits, symbols = block.compress(cur_rec, strided[idx: idx + block.pred_steps], self.bin2oct_kernel)
            pending.append(ops.logits_to_ranges(logits.contiguous(), PRE_SHIFT, symbols.contiguous()))
        sizes = [s.shape[0] for s, _ in pending]
        start_h = torch.empty(sum(sizes), dtype=torch.int16, pin_memory=True)
        freq_h = torch.empty(sum(sizes), dtype=torch.int16, pin_memory=True)
        start_h.copy_(torch.cat([s for s, _ in pending]), non_blocking=True)
        freq_h.copy_(torch.cat([f for _, f in pending]), non_blocking=True)
        bottom_h = torch.empty((top.C.shape[0], 3), dtype=torch.int32, pin_memory=True)
        bottom_h.copy_(top.C[:, 1:], non_blocking=True)
        offset_h = torch.empty((B, 3), dtype=offsets.dtype, pin_memory=True)
        offset_h.copy_(offsets[:, 1:], non_blocking=True)
        torch.cuda.current_stream().synchronize()

        start_np, freq_np = start_h.numpy().view(np.uint16), freq_h.numpy().view(np.uint16)
        rows = np.array([level_rows[idx] for idx in range(levels, 0, -1)], dtype=np.int64)     # [coded levels (coarse -> fine), B]
        if (rows.sum(1) != np.array(sizes)).any():
            raise RuntimeError('cloud row counts do not add up to the levels')
        level_start = np.concatenate(([0], np.cumsum(sizes)))
        cloud_start = np.concatenate((np.zeros((len(sizes), 1), np.int64), np.cumsum(rows, 1)), 1)
        bottom_np, offs = bottom_h.numpy(), offset_h.numpy()
        bottom_rows = rows[0]                                               # the coarsest coded level's rows ARE the bottom voxels
        bottom_at = np.concatenate(([0], np.cumsum(bottom_rows)))
        while len(self._extra_encoders) < B - 1:
            self._extra_encoders.append(RansEncoder(32 * 1024 * 1024))
        encoders = [self.rans_encoder, *self._extra_encoders[:B - 1]]

        def code_cloud(c: int) -> bytes:
            enc = encoders[c]
            for lvl in range(len(sizes) - 1, -1, -1):                       # finest level first: the decoder pops coarse -> fine
                a = int(level_start[lvl] + cloud_start[lvl, c])
                enc.encode_ranges(start_np[a: a + rows[lvl, c]], freq_np[a: a + rows[lvl, c]])
            mine = bottom_np[bottom_at[c]: bottom_at[c + 1]].reshape(-1)
            self.rans_encode_fea(self.bottom_cdf(mine), mine.astype(np.uint16), enc)
            head = b''.join(int(v).to_bytes(2, 'little') for v in offs[c].tolist()) + int(bottom_rows[c]).to_bytes(2, 'little')
            return head + enc.flush()
        return list(self._thread_pool(B).map(code_cloud, range(B)))

    @ops.no_gc_pause
    @torch.no_grad()
    def decompress_many(self, streams: List[bytes]) -> List[torch.Tensor]:
        """inverse of compress_many: the clouds' streams decoded in one traversal (per level one CDF launch and one set of copies for all
        clouds, the clouds' symbols decoded side by side on host threads); -> one int32 [n_b, 3] tensor per cloud"""
        B = len(streams)
        if B == 1:
            return [self.decompress(streams[0])]
        if B == 0:
            raise RuntimeError('decompress_many() takes a non-empty list of streams')
        if self.device_decoder:
            raise NotImplementedError('the device-side decoder takes one cloud')
        device = self.fold2bin_kernel.device
        skip = self.cfg.skip_top_scales_num
        blocks = self.blocks_dec[skip:]
        levels = self.max_downsample_times - skip
        decoders, bottoms, offsets = [], [], []
        for b, data in enumerate(streams):
            offsets.append([int.from_bytes(data[i:i + 2], 'little') for i in (0, 2, 4)])
            n_bottom = int.from_bytes(data[6:8], 'little')
            dec = RansDecoder()
            dec.flush(data[8:])
            xyz = torch.from_numpy(self.rans_decode_fea(n_bottom * 3, dec).astype(np.int32)).reshape(-1, 3)
            bottoms.append(F.pad(xyz, (1, 0, 0, 0), value=b))
            decoders.append(dec)
        self._clouds = {'decoders': decoders, 'rows': [b.shape[0] for b in bottoms], 'threads': self._thread_pool(B)}
        try:
            cur_rec = self.get_init_pc(torch.cat(bottoms).to(device), 2 ** levels)
            cur_bins, top_rec, top_stride, cur_bin = [], None, None, None
            for idx in range(levels, 0, -1):
                block = self._block(idx, blocks)
                if isinstance(block, OneScalePredictor):
                    cur_rec, cur_bin = block.decompress(cur_rec, self.bin2oct_kernel, self.unfold_kernel, self.rans_decode_oct,
                                                        if_upsample=idx != 1 and block.if_upsample,
                                                        next_block=self._block(idx - 1, blocks) if idx > 1 else None)
                else:
                    cur_bins.append(cur_bin)
                    cur_rec, cur_bin, top_rec, top_stride = block.decompress(
                        cur_rec, cur_bins, top_rec, top_stride, self.bin2oct_kernel, self.unfold_kernel, self.rans_decode_oct)
            if top_rec is None:
                if cur_rec.stride[0] != 2:
                    raise RuntimeError('unexpected final stride')
                parents = cur_rec.C
            else:
                if top_stride != 2:
                    raise RuntimeError('unexpected final stride')
                parents = top_rec
            recon = _children_of(parents, self.unfold_kernel, cur_bin)[:, 1:]
            counts = self._clouds['rows']                                   # after the last level: the clouds' point counts
        finally:
            self._clouds = None
        if sum(counts) != recon.shape[0]:
            raise RuntimeError('decoded point counts do not add up')
        offset_t = torch.tensor(offsets, device=device, dtype=torch.int32)
        out, at = [], 0
        for b, n in enumerate(counts):
            out.append(recon[at: at + n] + offset_t[b][None])
            at += n
        return out

    def compress_partitions(self, batched_coord: List[torch.Tensor]) -> bytes:
        parts = list(batched_coord[1:])                                      # element 0: the unpartitioned cloud
        coded: List[bytes] = []
        for g in self._groups([p.shape[0] for p in parts]):
            coded.extend(self.compress_many([parts[i] for i in g]))
        return b''.join(len(s).to_bytes(3, 'little') + s for s in coded)

    @ops.no_gc_pause
    @torch.no_grad()
    def decompress(self, compressed_bytes: bytes) -> torch.Tensor:
        device = self.fold2bin_kernel.device
        coord_offset = [int.from_bytes(compressed_bytes[i:i + 2], 'little') for i in (0, 2, 4)]
        n_bottom = int.from_bytes(compressed_bytes[6:8], 'little')
        payload = compressed_bytes[8:]
        self.rans_decoder.flush(payload)
        skip = self.cfg.skip_top_scales_num
        blocks = self.blocks_dec[skip:]
        levels = self.max_downsample_times - skip

        bottom = torch.from_numpy(self.rans_decode_fea(n_bottom * 3).astype(np.int32)).reshape(-1, 3)
        if self.device_decoder:
            # the few bottom symbols (65 K-entry side tables) are decoded on the host; the device continues from there
            x, pos = self.rans_decoder.tell()
            self._dev_stream, self._dev_stream_len = ops.stream_to_device(payload, device), len(payload)
            self._dev_state = torch.tensor([x - (1 << 32) if x >= 1 << 31 else x, pos & 0xffffffff if pos < 1 << 31 else pos - (1 << 32),
                                            pos >> 32, 0], dtype=torch.int32).to(device)
        cur_rec = self.get_init_pc(F.pad(bottom, (1, 0, 0, 0)).to(device), 2 ** levels)
        cur_bins, top_rec, top_stride, cur_bin = [], None, None, None
        for idx in range(levels, 0, -1):
            block = self._block(idx, blocks)
            if isinstance(block, OneScalePredictor):
                cur_rec, cur_bin = block.decompress(cur_rec, self.bin2oct_kernel, self.unfold_kernel, self.rans_decode_oct,
                                                    if_upsample=idx != 1 and block.if_upsample,
                                                    next_block=self._block(idx - 1, blocks) if idx > 1 else None)
            else:
                cur_bins.append(cur_bin)
                cur_rec, cur_bin, top_rec, top_stride = block.decompress(
                    cur_rec, cur_bins, top_rec, top_stride, self.bin2oct_kernel, self.unfold_kernel, self.rans_decode_oct)
        if top_rec is None:
            if cur_rec.stride[0] != 2:
                raise RuntimeError('unexpected final stride')
            parents = cur_rec.C
        else:
            if top_stride != 2:
                raise RuntimeError('unexpected final stride')
            parents = top_rec
        recon = _children_of(parents, self.unfold_kernel, cur_bin)[:, 1:]
        return recon + torch.tensor(coord_offset, device=device, dtype=torch.int32)[None]

    def decompress_partitions(self, concat_bytes: bytes) -> torch.Tensor:
        streams, pos = [], 0
        while pos != len(concat_bytes):
            length = int.from_bytes(concat_bytes[pos:pos + 3], 'little')
            streams.append(concat_bytes[pos + 3: pos + 3 + length])
            pos += 3 + length
        out: List[torch.Tensor] = []
        for g in self._groups([len(s) for s in streams]):                    # (bytes ~ voxels at a few bits per voxel: a size proxy)
            out.extend(self.decompress_many([streams[i] for i in g]))
        return torch.cat(out, 0)

    def test_forward(self, pc_data: PCData) -> dict:
        whole = isinstance(pc_data.xyz, torch.Tensor)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        data = self.compress(pc_data.xyz) if whole else self.compress_partitions(pc_data.xyz)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        recon = self.decompress(data) if whole else self.decompress_partitions(data)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        if pc_data.inv_transform is not None:
            inv = pc_data.inv_transform[0].to(recon.device)
            recon = recon * inv[3] + inv[None, :3]
            data = pc_data.inv_transform[0].numpy().astype('<f4').tobytes() + data
        n_org = pc_data.org_points_num[0] if pc_data.org_points_num else \
            (pc_data.xyz.shape[0] if whole else pc_data.xyz[0].shape[0])
        return {'pred': recon, 'compressed_bytes': data, 'bpp': 8 * len(data) / n_org, 'encode time': t1 - t0,
                'decode time': t2 - t1}
