"""TEST ORACLE -- not product code.

CPU restatement of the lossy_coord_v2 encode/decode path of the reference, driven by a plain dict of weights whose keys
are the reference's state_dict keys:

    Encoder / Decoder / get_keep            /root/reference/models/convolutional/lossy_coord_v2/layers.py:28-180
    EncoderGeoLossl, hyper decoders, ...    /root/reference/models/convolutional/lossy_coord_v2/layers.py:201-415
    GeoLosslessEntropyModel.compress/...    /root/reference/models/convolutional/lossy_coord_lossy_color/geo_lossl_em.py:59-317
    PCC.compress / decompress               /root/reference/models/convolutional/lossy_coord_v2/model.py:230-275
    conv / linear / PReLU blocks            /root/reference/lib/minkowski_sparse_conv_layers.py:31-159

Coordinates follow oracle/coords.py (MinkowskiEngine semantics restated), arithmetic oracle/sparse_conv.py, entropy
coding oracle/rans.py.  Parity: the rANS and framing parts are PINNED (golden vectors from the reference); the float
network part is UNPINNED against MinkowskiEngine -- what is checked is GPU == this oracle.

`conv='mm'` evaluates convolutions the reference-shaped way (gather, GEMM, scatter-add); `conv='chain'` uses the
fixed-order FMA chain with `order_fn(kind, c1, c2, c_out, n_out) -> 0|1|2` choosing the summation order per layer (see
oracle/sparse_conv.c), which makes the activations comparable bit for bit with a device kernel of the same order.
"""
import io
import math
from typing import Callable, Dict, List, Optional, Tuple

import numpy as np
import torch

from . import coords as oc
from . import sparse_conv as sc
from .rans import BinaryRansCoder, IndexedRansCoder


# ---- framing (restated from the format) -----------------------------------------------------
def _u(x: int, n: int) -> bytes:
    return int(x).to_bytes(n, 'little', signed=False)


def _r(bs: io.BytesIO, n: int) -> int:
    return int.from_bytes(bs.read(n), 'little', signed=False)


def concat_strings(strings: List[bytes]) -> bytes:
    """header ('1' marker + (width-1) per string on 1 or 2 bits, big-endian, ceil(n/items_per_byte + 0.25) bytes, bit 7
    of byte 0 set in 2-bit mode) | lengths LE | payloads
    (/root/reference/lib/entropy_models/hyperprior/noisy_deep_factorized/utils.py:8-44)"""
    widths = [max(1, math.ceil(len(s).bit_length() / 8)) for s in strings]
    bits = max(1, max((w - 1).bit_length() for w in widths))
    text = '1' + ''.join(format(w - 1, f'0{bits}b') for w in widths)
    head = int(text, 2).to_bytes(math.ceil(len(strings) / (8 // bits) + 0.25), 'big')
    if bits == 2:
        head = bytes([head[0] | 0x80]) + head[1:]
    return head + b''.join(len(s).to_bytes(w, 'little') for s, w in zip(strings, widths)) + b''.join(strings)


def split_strings(bs: io.BytesIO, count: int) -> List[bytes]:
    first = bs.read(1)[0]
    bits = 2 if first & 0x80 else 1
    n_head = math.ceil(count / (8 // bits) + 0.25)
    text = f"{int.from_bytes(bytes([first & 0x7f]) + bs.read(n_head - 1), 'big'):b}"[1:]
    widths = [int(text[i: i + bits], 2) + 1 for i in range(0, count * bits, bits)]
    lengths = [int.from_bytes(bs.read(w), 'little') for w in widths]
    return [bs.read(n) for n in lengths]


class Feature:
    """features [n, C] (torch fp32) on an oracle coordinate level"""

    def __init__(self, f: torch.Tensor, level: oc.Level):
        assert f.shape[0] == level.n
        self.f, self.level = f, level


class OracleV2:
    def __init__(self, weights: Dict[str, torch.Tensor], cfg, conv: str = 'mm',
                 order_fn: Optional[Callable[[str, int, int, int, int], int]] = None):
        self.P = {k: v.detach().cpu().float() for k, v in weights.items() if isinstance(v, torch.Tensor)}
        self.cfg = cfg
        self.conv = conv
        self.order_fn = order_fn or (lambda kind, c1, c2, c_out, n_out: 0)
        self._kmaps = {}
        self.trace: Dict[str, np.ndarray] = {}     # activations by layer prefix (filled when keep_trace)
        # The reference goes on evaluating the feature predictors below the last coded level and discards the result
        # (geo_lossl_em.py:196-210, `del lower_fea_recon`).  True stops where the product stops (same bytes: nothing reads
        # that tail); used when this oracle is TIMED beside the GPU path, so that both do the same work.
        self.skip_unused_tail = False
        self.keep_trace = False

    # ---- primitive layers ----------------------------------------------------------------------------------------
    def _slope(self, key: str) -> Tuple[int, float]:
        if key in self.P:
            return sc.ACT_PRELU, float(self.P[key].reshape(-1)[0])
        return sc.ACT_NONE, 0.0

    def _apply(self, name, kind, x1, x2, kmap, n_out, w, b, act, slope, clip):
        c1 = x1.shape[1]
        c2 = 0 if x2 is None else x2.shape[1]
        c_out = w.shape[-1]
        if self.conv == 'mm':
            x = x1 if x2 is None else torch.cat((x1, x2), 1)
            out = sc.conv_mm(x, kmap, w, b, n_out, act, slope, clip)
        else:
            table = oc.dense_table(kmap, n_out)
            out = torch.from_numpy(sc.conv_chain(x1.numpy(), table, w.numpy(), None if b is None else b.numpy(), n_out,
                                                 x2=None if x2 is None else x2.numpy(), act=act, slope=slope,
                                                 clip=clip, order=self.order_fn(kind, c1, c2, c_out, n_out)))
        if self.keep_trace:
            self.trace[name] = out.numpy().copy()
        return out

    def _kmap(self, src: oc.Level, dst: oc.Level, kind: str):
        key = (id(src), id(dst), kind)
        if key not in self._kmaps:
            if kind == 'k3':
                m = oc.kernel_map(src, dst, 3)
            elif kind == 'k2s2':
                m = oc.kernel_map(src, dst, 2)
            elif kind == 'k2s2T':
                m = oc.transposed_map(src, dst)
            else:
                raise ValueError(kind)
            self._kmaps[key] = (m, src, dst)     # keep the levels alive so ids stay unique
        return self._kmaps[key][0]

    def conv_block(self, prefix: str, x: Feature, kind: str, dst: Optional[oc.Level] = None, x2: Optional[Feature] = None,
                   clip: float = 0.0) -> Feature:
        """kind: 'k1' | 'k3' | 'k2s2' | 'k2s2T' (needs dst) | 'gen'"""
        w = self.P[prefix + '.conv.kernel']
        b = self.P.get(prefix + '.conv.bias')
        b = None if b is None else b.reshape(-1)
        act, slope = self._slope(prefix + '.act_module.module.weight')
        src = x.level
        if kind == 'k1':
            dst = src
            kmap = [(np.arange(src.n), np.arange(src.n))]
            w = w.reshape(1, *w.shape[-2:])
        elif kind == 'k3':
            dst = src
            kmap = self._kmap(src, dst, 'k3')
        elif kind == 'k2s2':
            dst = dst or self._strided(src)
            kmap = self._kmap(src, dst, 'k2s2')
        elif kind == 'k2s2T':
            kmap = self._kmap(src, dst, 'k2s2T')
        elif kind == 'gen':
            dst = oc.generated(src)
            kmap = self._kmap(src, dst, 'k2s2T')
        else:
            raise ValueError(kind)
        out = self._apply(prefix, kind, x.f, None if x2 is None else x2.f, kmap, dst.n, w, b, act, slope, clip)
        return Feature(out, dst)

    def _strided(self, src: oc.Level) -> oc.Level:
        # one coarser level per source level, like a coordinate manager's cached stride map
        if not hasattr(src, '_coarser'):
            src._coarser = oc.strided(src)
        return src._coarser

    def mlp_block(self, prefix: str, x: Feature, x2: Optional[Feature] = None, clip: float = 0.0) -> Feature:
        weight = self.P[prefix + '.mlp.linear.weight']                    # [out, in]
        b = self.P.get(prefix + '.mlp.linear.bias')
        act, slope = self._slope(prefix + '.act.module.weight')
        n = x.level.n
        if self.conv == 'mm':
            # MinkowskiLinear is torch's linear on the feature matrix (lib/minkowski_sparse_conv_layers.py:38,43)
            f = x.f if x2 is None else torch.cat((x.f, x2.f), 1)
            out = sc._act(torch.nn.functional.linear(f, weight, b), act, slope, clip)
            if self.keep_trace:
                self.trace[prefix] = out.numpy().copy()
            return Feature(out, x.level)
        w = weight.t().contiguous()                                       # [in, out]
        kmap = [(np.arange(n), np.arange(n))]
        out = self._apply(prefix, 'mlp', x.f, None if x2 is None else x2.f, kmap, n, w.reshape(1, *w.shape), b, act, slope, clip)
        return Feature(out, x.level)

    # ---- networks ------------------------------------------------------------------------------------------------
    def encoder(self, x: Feature):
        counts = []
        n_blocks = len(self.cfg.encoder_channels)
        x = self.conv_block('encoder.blocks.0', x, 'k3')
        if n_blocks > 1:
            counts.append([x.level.n])
        for i in range(1, n_blocks):
            x = self.conv_block(f'encoder.blocks.{i}.0', x, 'k2s2')
            x = self.conv_block(f'encoder.blocks.{i}.1', x, 'k3')
            if i != n_blocks - 1:
                counts.append([x.level.n])
        counts = [[int(n * self.cfg.adaptive_pruning_scaler) for n in c] for c in counts]
        return x, counts

    def em_encoder(self, x: Feature) -> List[Feature]:
        pre = 'em_lossless_based.encoder'
        cfg = self.cfg
        outs = [self.mlp_block(pre + '.blocks_out_first', x) if cfg.skip_encoding_fea < 0 else x]
        n = len(cfg.geo_lossl_if_sample)
        bound = float(cfg.bottleneck_value_bound)
        for i, down in enumerate(cfg.geo_lossl_if_sample):
            x = self.conv_block(f'{pre}.blocks.{i}.0', x, 'k2s2' if down else 'k3')
            x = self.conv_block(f'{pre}.blocks.{i}.1', x, 'k3')
            if i >= cfg.skip_encoding_fea:
                outs.append(self.mlp_block(f'{pre}.blocks_out.{i}', x, clip=bound if i == n - 1 else 0.0))
            else:
                outs.append(x)
        return outs

    def hyper_coord(self, idx: int, lower: Feature) -> Feature:
        pre = f'em_lossless_based.hyper_decoder_coord.blocks.{idx}'
        g = self.conv_block(pre + '.0', lower, 'gen')
        return self.conv_block(pre + '.1', g, 'k3')

    def hyper_fea(self, idx: int, lower: Feature, target: oc.Level) -> Feature:
        pre = f'em_lossless_based.hyper_decoder_fea.blocks.{idx}'
        if self.cfg.geo_lossl_if_sample[idx]:
            x = self.conv_block(pre + '.0', lower, 'k2s2T', dst=target)
        else:
            x = self.conv_block(pre + '.0', lower, 'k3')
        return self.conv_block(pre + '.1', x, 'k3')

    def residual(self, idx: int, fea: Feature, pred: Feature) -> Feature:
        pre = f'em_lossless_based.residual_block.blocks.{idx}.blocks'
        x = self.conv_block(pre + '.0', fea, 'k3', x2=pred)
        return self.conv_block(pre + '.1', x, 'k3', clip=float(self.cfg.bottleneck_value_bound))

    def em_decoder_block(self, idx: int, pred: Feature, res: Optional[torch.Tensor] = None) -> Feature:
        pre = f'em_lossless_based.decoder_block.blocks.{idx}'
        if idx > self.cfg.skip_encoding_fea:
            r = Feature(res, pred.level)
            r = self.mlp_block(pre + '.residual_decoder.0', r)
            r = self.mlp_block(pre + '.residual_decoder.1', r)
            x = self.mlp_block(pre + '.decoder.0', r, x2=pred)
            return self.mlp_block(pre + '.decoder.1', x)
        x = self.mlp_block(pre + '.decoder.0', pred)
        return self.mlp_block(pre + '.decoder.1', x)

    # ---- entropy coding helpers ----------------------------------------------------------------------------------
    def init_prob(self, logit: torch.Tensor) -> np.ndarray:
        """geo_lossl_em.py:96-99.  conv == 'mm' (the reference-shaped evaluation): torch.sigmoid as the reference calls it; 'chain'
        (the evaluation that models the HIP path bit for bit): the logistic function the HIP path specifies (sparse_conv.sigmoid_spec)"""
        s = torch.sigmoid(logit).numpy() if getattr(self, 'conv', 'mm') == 'mm' else sc.sigmoid_spec(logit)
        return np.clip(np.round(s.astype(np.float64) * (1 << 16)).astype(np.uint32), 1, (1 << 16) - 1)

    @staticmethod
    def rans_encode_with_cdf(target: np.ndarray, bs: io.BytesIO, offset: Optional[int] = None):
        coder = IndexedRansCoder(False, 1)
        bs.write(_u(target.shape[0], 3))
        if offset is None:
            offset = int(target.min())
            bs.write(_u(-offset, 1))
        pmf = np.bincount((target - offset).reshape(-1)).astype(np.float64)
        coder.init_with_pmfs(pmf[None], np.array([offset], dtype=np.int32))
        cdf = coder.get_cdfs()[0]
        bs.write(_u(len(cdf) - 2, 1))
        for c in cdf[1:-1]:
            bs.write(_u(c, 2))
        payload = coder.encode(target.reshape(1, -1).astype(np.int32))[0]
        bs.write(_u(len(payload), 3))
        bs.write(payload)

    @staticmethod
    def rans_decode_with_cdf(bs: io.BytesIO, offset: Optional[int] = None, channels: int = 1) -> np.ndarray:
        coder = IndexedRansCoder(False, 1)
        rows = _r(bs, 3)
        if offset is None:
            offset = -_r(bs, 1)
        cdf = [0, *(_r(bs, 2) for _ in range(_r(bs, 1))), 1 << 16]
        coder.init_with_quantized_cdfs([cdf], np.array([offset], dtype=np.int32))
        payload = bs.read(_r(bs, 3))
        out = np.empty((1, rows * channels), np.int32)
        coder.decode([payload], out)
        return out.reshape(rows, channels)

    # ---- lossless entropy model ----------------------------------------------------------------------------------
    def em_compress(self, y_top: Feature) -> bytes:
        cfg = self.cfg
        scaler = float(cfg.bottleneck_scaler)
        *feas, bottom = self.em_encoder(y_top)
        bottom.f = torch.round(bottom.f * scaler)
        res_list = [bottom.f.to(torch.int32).numpy()]
        bottom.f = bottom.f / scaler
        lower = bottom
        coord_strings = []
        self.symbols = {'occupancy': [], 'prob': []}
        strides = [f.level.stride for f in feas] + [bottom.level.stride]
        last_coded = min((i for i in range(len(feas)) if i > cfg.skip_encoding_fea or strides[i] != strides[i + 1]),
                         default=len(feas))
        for idx in range(len(feas) - 1, -1, -1):
            fea = feas[idx]
            target = fea.level
            if lower.level.stride != target.stride:
                pred = self.hyper_coord(idx, lower)
                mask = target.rows_of(pred.level.coords) >= 0
                prob = self.init_prob(pred.f)
                coord_strings.append(BinaryRansCoder(1).encode(mask.reshape(1, -1), prob.reshape(1, -1))[0])
                self.symbols['occupancy'].append(mask)
                self.symbols['prob'].append(prob.reshape(-1))
            if self.skip_unused_tail and idx <= last_coded and idx <= cfg.skip_encoding_fea:
                break
            fea_pred = self.hyper_fea(idx, lower, target)
            if idx > cfg.skip_encoding_fea:
                res = self.residual(idx, fea, fea_pred).f
                res = torch.round(res * scaler)
                res_list.append(res.to(torch.int32).numpy())
                res = res / scaler
                if self.skip_unused_tail and idx == last_coded:
                    break
                lower = self.em_decoder_block(idx, fea_pred, res)
            else:
                lower = self.em_decoder_block(idx, fea_pred)
        self.symbols['residual'] = np.concatenate(res_list, 0)
        with io.BytesIO() as bs:
            bs.write(_u(int(math.log2(bottom.level.stride)), 1))
            bs.write(_u(bottom.level.n, 3))
            self.rans_encode_with_cdf(self.symbols['residual'], bs)
            bs.write(_u(len(coord_strings), 1))
            bs.write(concat_strings(coord_strings))
            self.rans_encode_with_cdf((bottom.level.coords[:, 1:] // bottom.level.stride).astype(np.int32), bs, 0)
            return bs.getvalue()

    def em_decompress(self, data: bytes) -> Feature:
        cfg = self.cfg
        scaler = float(cfg.bottleneck_scaler)
        with io.BytesIO(data) as bs:
            bottom_stride = 2 ** _r(bs, 1)
            n_bottom = _r(bs, 3)
            res_all = torch.from_numpy(self.rans_decode_with_cdf(bs).astype(np.float32)) / scaler
            n_strings = _r(bs, 1)
            coord_strings = split_strings(bs, n_strings)
            bottom_xyz = self.rans_decode_with_cdf(bs, 0, 3) * bottom_stride
        coords = np.concatenate((np.zeros((n_bottom, 1), np.int64), bottom_xyz.astype(np.int64)), 1)
        level = oc.Level(coords, bottom_stride)
        assert (level.order == np.arange(level.n)).all(), 'bottom coordinates are expected in Morton order'
        lower = Feature(res_all[:n_bottom], level)
        used = n_bottom
        cur = level
        for idx in range(len(cfg.geo_lossl_if_sample) - 1, -1, -1):
            if cfg.geo_lossl_if_sample[idx]:
                pred = self.hyper_coord(idx, lower)
                prob = self.init_prob(pred.f)
                bits = np.empty((1, prob.size), dtype=np.bool_)
                BinaryRansCoder(1).decode([coord_strings.pop(0)], prob.reshape(1, -1), bits)
                cur = oc.Level(pred.level.coords[bits.reshape(-1)], pred.level.stride)
            fea_pred = self.hyper_fea(idx, lower, cur)
            if idx > cfg.skip_encoding_fea:
                res = res_all[used: used + cur.n]
                used += cur.n
                lower = self.em_decoder_block(idx, fea_pred, res)
            else:
                lower = self.em_decoder_block(idx, fea_pred)
        assert not coord_strings and used == res_all.shape[0]
        return lower

    # ---- lossy decoder -------------------------------------------------------------------------------------------
    def get_keep(self, pred: Feature, parent: oc.Level, target: Optional[int]) -> np.ndarray:
        """layers.py:151-180 for one sample: max-pool onto the cells of `parent` -- the decoder's input level, tensor
        stride 2^stages (max_stride_lossy_recon) --, un-pool, k-th value threshold."""
        v = pred.f.reshape(-1).numpy()
        q = pred.level.coords.copy()
        q[:, 1:] = q[:, 1:] // parent.stride * parent.stride
        cell = parent.rows_of(q)
        cell_max = np.full(parent.n, -np.inf, dtype=np.float32)
        np.maximum.at(cell_max, cell, v)
        not_max = (v - cell_max[cell]) != 0
        if target is None:                       # adaptive_pruning = False: fixed threshold 0 (layers.py:176-180)
            return (v > 0) | ~not_max
        ranked = np.sort(v[not_max])
        kth = v.shape[0] - target
        assert v.shape[0] > target and 1 <= kth <= ranked.shape[0]
        thr = ranked[kth - 1]
        return (v > thr) | ~not_max

    def decoder(self, fea: Feature, points_num_list: List[List[int]]) -> np.ndarray:
        n_stage = len(self.cfg.decoder_channels)
        parent = fea.level                      # every stage takes its local maxima inside the voxels of this level
        for i in range(n_stage):
            up = f'decoder.upsample_blocks.{i}'
            j = 0
            if i == n_stage - 1:
                fea = self.conv_block(f'{up}.{j}', fea, 'k3')
                j += 1
            fea = self.conv_block(f'{up}.{j}', fea, 'gen')
            j += 1
            if i != n_stage - 1:
                fea = self.conv_block(f'{up}.{j}', fea, 'k3')
            pred = self.conv_block(f'decoder.classify_blocks.{i}.0', fea, 'k1')
            pred = self.conv_block(f'decoder.classify_blocks.{i}.1', pred, 'k1')
            keep = self.get_keep(pred, parent, None if points_num_list is None else points_num_list.pop()[0])
            if i != n_stage - 1:
                lvl = oc.Level(fea.level.coords[keep], fea.level.stride)
                fea = Feature(fea.f[torch.from_numpy(keep)], lvl)
            else:
                return fea.level.coords[keep][:, 1:]

    # ---- frame ---------------------------------------------------------------------------------------------------
    def compress(self, batched_coord: np.ndarray) -> bytes:
        c = np.asarray(batched_coord, dtype=np.int64)
        offset = c[:, 1:].min(0)
        c = c.copy()
        c[:, 1:] -= offset
        level = oc.Level(c, 1)
        x = Feature(torch.ones((level.n, 1), dtype=torch.float32), level)
        fea, counts = self.encoder(x)
        em = self.em_compress(fea)
        out = b''.join(_u(v, 2) for v in offset.tolist())
        if self.cfg.adaptive_pruning:
            out += b''.join(_u(cnt[0], 3) for cnt in counts)
        return out + em

    def decompress(self, data: bytes) -> np.ndarray:
        with io.BytesIO(data) as bs:
            offset = [_r(bs, 2) for _ in range(3)]
            counts = [[_r(bs, 3)] for _ in range(len(self.cfg.encoder_channels) - 1)] if self.cfg.adaptive_pruning else None
            em = bs.read()
        fea = self.em_decompress(em)
        xyz = self.decoder(fea, counts)
        return (xyz + np.array(offset, dtype=np.int64)).astype(np.int32)
