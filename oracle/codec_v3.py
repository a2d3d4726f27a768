"""TEST ORACLE -- not product code.

CPU restatement of the test-time path of the reference's octree codec with coded latents, driven by a dict of tensors with
the reference's state_dict keys:

    OneScalePredictor.compress / decompress                /root/reference/models/convolutional/lossy_coord_v3/model.py:171-245
    Model.compress / decompress, header                    /root/reference/models/convolutional/lossy_coord_v3/model.py:547-681
    batch_quantize_pmf_torch, rans_encode_fea / decode_fea /root/reference/models/convolutional/lossy_coord_v3/model.py:501-545
    encoder stacks, Fold, Block, SparseSequential          /root/reference/models/convolutional/lossy_coord_v3/model.py:248-267,350-365,684-710
    kernel-offset enumeration (torchsparse == the reference's own integer engine, cuda_ops.py:257-260)
                                                           oracle/codec_int.py:kernel_table
    rANS stream                                            oracle/rans.c (pinned by the reference's golden streams)

Parity: the framing, side information and histogram CDFs are exact integer / float32 arithmetic stated by the reference;
the float network is UNPINNED against torchsparse (not installable here).  `conv='mm'` evaluates a convolution per kernel
offset as gather -> GEMM -> accumulate; `conv='chain'` as one fixed-order FMA chain per output element in the order
`order_fn(c_in, c_out, n_offsets, n_out)` names (oracle/sparse_conv.c), so a device kernel documenting the same order can be
compared bit for bit.
"""
from typing import Callable, Dict, List, Optional

import numpy as np
import torch

from . import sparse_conv as sc
from .codec_int import Sp, _np, kernel_table
from .coords import morton_encode
from .rans import RansDecoder, RansEncoder


def prelu(x: np.ndarray, slope: np.ndarray) -> np.ndarray:
    return np.where(x > 0, x, x * slope.astype(np.float32)).astype(np.float32)


def quantize_pmf(p: np.ndarray, softmax: bool) -> np.ndarray:
    """rows of logits / probabilities -> uint16 CDF rows without the leading zero (:501-509), float32 throughout"""
    t = torch.from_numpy(np.ascontiguousarray(p, dtype=np.float32))
    if softmax:
        t = torch.softmax(t, dim=-1)
    t = t.mul(65536 - t.shape[1]).floor_().add_(1)
    t.cumsum_(-1)
    t[:, -1] = 65535
    return t.numpy().astype(np.uint16)


def histogram_cdf(values: np.ndarray) -> np.ndarray:
    counts = np.bincount(values.astype(np.int64), minlength=2)
    pmf = counts.astype(np.float32) / np.float32(values.size)
    return quantize_pmf(pmf[None], False)[0]


def top_children(logits: np.ndarray, points_num: int) -> np.ndarray:
    """every row's maximum, and everything above the (8n - points_num)-th smallest logit of the level (:216-221)"""
    mask = logits == logits.max(1, keepdims=True)
    k = logits.size - int(points_num)
    if k >= 1:
        kth = np.partition(logits.reshape(-1), k - 1)[k - 1]
        mask |= logits > kth
    else:
        mask |= True
    return mask


class OracleV3:
    def __init__(self, weights: Dict[str, object], cfg, conv: str = 'mm',
                 order_fn: Optional[Callable[[int, int, int, int], int]] = None):
        self.P = {k: _np(v).astype(np.float32) for k, v in weights.items()}
        self.cfg = cfg
        self.conv = conv
        self.order_fn = order_fn or (lambda c_in, c_out, n_offsets, n_out: 0)
        self.levels = int(np.log2(cfg.max_stride))
        self.bin2oct = np.arange(7, -1, -1)
        self.unfold = np.array([(0, dx, dy, dz) for dx in (0, 1) for dy in (0, 1) for dz in (0, 1)], dtype=np.int64)[None]
        self.cdf1 = np.arange(2, 65537, dtype=np.int64).astype(np.uint16)[None].copy()
        self.cdf2 = (np.arange(1, 129, dtype=np.int64) * 512).astype(np.uint16)[None].copy()
        self.cdf1[:, -1] = 65535
        self.cdf2[:, -1] = 65535
        self.n_enc = 0
        for idx in range(len(cfg.num_latents)):
            if all(v == 0 for v in cfg.num_latents[idx:]):
                break
            self.n_enc += 1
        self.trace: Dict[str, np.ndarray] = {}

    # -- layers --------------------------------------------------------------------------------------------------------
    def _gemm(self, x: np.ndarray, table: Optional[np.ndarray], w: np.ndarray, b: Optional[np.ndarray], n_out: int) -> np.ndarray:
        k, c_in, c_out = w.shape
        if self.conv == 'chain':
            return sc.conv_chain(x, None if table is None else table.astype(np.int32), w, b, n_out,
                                 order=self.order_fn(c_in, c_out, k, n_out))
        # gather -> GEMM -> scatter-add per kernel offset, bias last: the structure (and, on one machine, the bits) of a
        # torch evaluation of the same sum
        xt, wt = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)), torch.from_numpy(np.ascontiguousarray(w))
        out = torch.zeros((n_out, c_out), dtype=torch.float32)
        for i in range(k):
            rows = torch.from_numpy(np.nonzero(table[i] >= 0)[0])
            if len(rows):
                out.index_add_(0, rows, torch.mm(xt.index_select(0, torch.from_numpy(table[i][rows.numpy()].astype(np.int64))), wt[i]))
        if b is not None:
            out += torch.from_numpy(b)
        return out.numpy()

    def sconv(self, pre: str, x: Sp, ks: int = 3, st: int = 1) -> Sp:
        w = self.P[pre + '.kernel']
        w = w.reshape(-1, w.shape[-2], w.shape[-1])
        if st == 1:
            out_c, out_stride = x.C, x.stride
        else:
            out_stride = x.stride * st
            out_c = x.caches['cmaps'][out_stride]
        tag = (x.stride, ks, st)
        if tag not in x.caches['tables']:
            x.caches['tables'][tag] = kernel_table(x.C, out_c, (ks,) * 3, (st,) * 3)
        f = self._gemm(x.F, x.caches['tables'][tag], w, self.P.get(pre + '.bias'), len(out_c))
        return Sp(f, out_c, out_stride, x.caches)

    def linear(self, pre: str, f: np.ndarray) -> np.ndarray:
        w, b = self.P[pre + '.weight'], self.P.get(pre + '.bias')      # [out, in]
        if self.conv == 'chain':
            return sc.conv_chain(f, None, np.ascontiguousarray(w.T)[None], b, f.shape[0],
                                 order=self.order_fn(w.shape[1], w.shape[0], 1, f.shape[0]))
        return torch.nn.functional.linear(torch.from_numpy(np.ascontiguousarray(f, dtype=np.float32)), torch.from_numpy(w),
                                          None if b is None else torch.from_numpy(b)).numpy()

    def act(self, pre: str, f: np.ndarray) -> np.ndarray:
        return prelu(f, self.P[pre + '.weight'])

    def block(self, pre: str, x: Sp) -> Sp:
        y = self.sconv(pre + '.conv', x)
        y.F = self.act(pre + '.act', y.F)
        y = self.sconv(pre + '.conv2', y)
        y.F = self.act(pre + '.act2', (y.F + x.F).astype(np.float32))
        return y

    # -- predictor pieces ----------------------------------------------------------------------------------------------
    def trunk(self, pre: str, x: Sp) -> Sp:
        if x.F.shape[1] == 1:
            x = self.sconv(pre + '.dec_init', x)
        return self.block(pre + '.dec', x)

    def to_latent(self, t: str, rec: Sp, ref: Sp) -> np.ndarray:
        a = self.act(t + '.0.1', self.linear(t + '.0.0', ref.F))
        h = self.act(t + '.1.1', self.linear(t + '.1.0', np.concatenate((a, rec.F), 1)))
        y = self.sconv(t + '.1.2', Sp(h, rec.C, rec.stride, rec.caches))
        y.F = self.act(t + '.1.3', y.F)
        return np.round(self.sconv(t + '.1.4', y).F)

    def absorb(self, t: str, rec: Sp, latent: np.ndarray) -> Sp:
        wide = self.act(t + '.2.1', self.linear(t + '.2.0', latent))
        h = self.act(t + '.3.1', self.linear(t + '.3.0', np.concatenate((rec.F, wide), 1)))
        return self.block(t + '.3.2', Sp(h, rec.C, rec.stride, rec.caches))

    def predict(self, pre: str, rec: Sp, lossless: bool) -> np.ndarray:
        y = self.sconv(pre + '.pred.0', rec)
        y.F = self.act(pre + '.pred.1', y.F)
        return self.linear(pre + '.pred.2', y.F) if lossless else self.sconv(pre + '.pred.2', y).F

    def expand(self, pre: str, rec: Sp, bits: np.ndarray, child_coords: np.ndarray, caches) -> Sp:
        u = pre + '.upsample'
        h = self.act(u + '.1', self.linear(u + '.0', np.concatenate((rec.F, bits.astype(np.float32)), 1)))
        y = self.block(u + '.2', Sp(h, rec.C, rec.stride, rec.caches))
        f = self.linear(u + '.3', y.F)
        f = f.reshape(f.shape[0], 8, f.shape[1] // 8)[bits.astype(bool)]
        return Sp(f, child_coords, rec.stride // 2, caches)

    def _block(self, idx: int, skip: int):
        """-> (state_dict prefix, number of latents, lossless?)"""
        n = len(self.cfg.num_latents) - skip
        if idx > n:
            return 'block_dec_recurrent', 0, True
        j = idx - 1 + skip
        return f'blocks_dec.{j}', self.cfg.num_latents[j], bool(self.cfg.lossl_geo_upsample[j])

    def _symbols(self, bits):
        return ((bits.astype(np.int64) << self.bin2oct).sum(1) - 1).astype(np.uint16)

    def _bits(self, symbols):
        return (((symbols.astype(np.int64)[:, None] + 1) >> self.bin2oct) & 1).astype(bool)

    def _children(self, coords, mask):
        c = coords.astype(np.int64)[:, None].copy()
        c[..., 1:] <<= 1
        return (c + self.unfold)[mask]

    # -- side information ----------------------------------------------------------------------------------------------
    def encode_fea(self, enc: RansEncoder, cdf: np.ndarray, values: np.ndarray, lo: Optional[int] = None):
        enc.encode(cdf[None], values.astype(np.uint16))
        enc.encode(self.cdf1, (cdf[:-1] - 1).astype(np.uint16))
        assert len(cdf) - 2 < 128
        enc.encode(self.cdf2, np.array([len(cdf) - 2], dtype=np.uint16))
        if lo is not None:
            assert 0 <= lo < 128
            enc.encode(self.cdf2, np.array([lo], dtype=np.uint16))

    def decode_fea(self, dec: RansDecoder, length: int, with_lo: bool = True) -> np.ndarray:
        lo = np.zeros(1, np.uint16)
        if with_lo:
            dec.decode(self.cdf2, lo)
        cdf_len = np.empty(1, np.uint16)
        dec.decode(self.cdf2, cdf_len)
        cdf = np.empty(int(cdf_len[0]) + 1, np.uint16)
        dec.decode(self.cdf1, cdf)
        cdf = np.pad(cdf + 1, (0, 1))
        cdf[-1] = 65535
        out = np.empty(length, np.uint16)
        dec.decode(cdf[None], out)
        return out.astype(np.int64) - int(lo[0])

    # -- codec ---------------------------------------------------------------------------------------------------------
    def compress(self, xyz: np.ndarray) -> bytes:
        cfg = self.cfg
        xyz = np.asarray(xyz, dtype=np.int64)
        offset = xyz[:, 1:].min(0)
        xyz = xyz.copy()
        xyz[:, 1:] -= offset
        xyz = xyz[np.argsort(morton_encode(xyz[:, 1:], 'xyz', inverse=True), kind='stable')]
        skip = cfg.skip_top_scales_num
        levels = self.levels - skip
        caches = {'cmaps': {}, 'tables': {}}
        coords, bits = [xyz], [None]
        for l in range(levels):
            c = coords[-1].copy()
            c[:, 1:] >>= 1
            keep = np.ones(len(c), bool)
            keep[1:] = (c[1:] != c[:-1]).any(1)
            parents = c[keep]
            table = kernel_table(coords[-1], parents, (2, 2, 2), (2, 2, 2))
            caches['tables'][(1 << l, 2, 2)] = table
            bits.append((table >= 0).T.astype(np.float32))
            coords.append(parents)
        for l, c in enumerate(coords):
            caches['cmaps'][1 << l] = c

        # encoder features of the levels that carry latents: blocks_enc[0] is the fold (the occupancy bits themselves)
        ref: List[Optional[Sp]] = [None] * (levels + 1)
        n_enc = max(self.n_enc - skip, 0)
        if n_enc:
            ref[1] = Sp(bits[1], coords[1], 2, caches)
        for i in range(1, n_enc):
            pre = f'blocks_enc.{i + skip}'
            x = ref[i]
            if i + skip == 1:
                x = self.sconv(pre + '.0', x)
                x.F = self.act(pre + '.1', x.F)
                x = self.sconv(pre + '.2', x, 2, 2)
                nxt = 3
                if cfg.channels >= 256:
                    x.F = self.act(pre + '.3', x.F)
                    nxt = 4
                x = self.block(f'{pre}.{nxt}', x)
            else:
                x = self.block(pre + '.1', self.sconv(pre + '.0', x, 2, 2))
            ref[i + 1] = x
            self.trace[f'enc{i + 1}'] = x.F

        n_lossy = next((i for i, v in enumerate(cfg.lossl_geo_upsample) if v == 1), len(cfg.lossl_geo_upsample))
        points_num = [len(coords[i]) for i in range(n_lossy)]
        cur = Sp(np.ones((len(coords[-1]), 1), np.float32), coords[-1], 1 << levels, caches)
        pending = []
        for idx in range(levels, 0, -1):
            pre, n_lat, lossless = self._block(idx, skip)
            if not lossless:
                assert n_lat == 0, 'latents on a lossy level are never written (model.py:590-592)'
                break
            cur = self.trunk(pre, cur)
            latents = []
            for j in range(n_lat):
                t = f'{pre}.transforms.{j}'
                z = self.to_latent(t, cur, ref[idx])
                self.trace[f'latent{idx}.{j}'] = z
                cur = self.absorb(t, cur, z)
                lo = int(-z.min())
                v = (z + lo).reshape(-1).astype(np.int64)
                latents.append((histogram_cdf(v), v, lo))
            logits = self.predict(pre, cur, True)
            self.trace[f'logits{idx}'] = logits
            sym = self._symbols(bits[idx])
            self.trace[f'symbols{idx}'] = sym
            pending.append((latents, quantize_pmf(logits, True), sym))
            if idx != 1:
                cur = self.expand(pre, cur, bits[idx], coords[idx - 1], caches)
        enc = RansEncoder(32 << 20)
        self.coded = {'symbols': [], 'rows': [], 'fea': []}          # what went to the coder, in coding order
        while pending:
            latents, rows, sym = pending.pop()
            enc.encode(rows, sym)
            self.coded['symbols'].append(sym)
            self.coded['rows'].append(rows)
            while latents:
                self.coded['fea'].append(latents[-1])
                self.encode_fea(enc, *latents.pop())
        bottom = coords[-1][:, 1:].reshape(-1)
        self.coded['fea'].append((histogram_cdf(bottom), bottom, None))
        self.encode_fea(enc, histogram_cdf(bottom), bottom)
        head = b''.join(int(v).to_bytes(2, 'little') for v in offset.tolist()) + (len(bottom) // 3).to_bytes(2, 'little')
        head += b''.join(int(n).to_bytes(3, 'little') for n in points_num)
        return head + enc.flush()

    def decompress(self, data: bytes) -> np.ndarray:
        cfg = self.cfg
        offset = np.array([int.from_bytes(data[i:i + 2], 'little') for i in (0, 2, 4)], dtype=np.int64)
        n_bottom = int.from_bytes(data[6:8], 'little')
        pos, points_num = 8, []
        for v in cfg.lossl_geo_upsample:
            if v == 1:
                break
            points_num.append(int.from_bytes(data[pos:pos + 3], 'little'))
            pos += 3
        dec = RansDecoder()
        dec.flush(data[pos:])
        skip = cfg.skip_top_scales_num
        levels = self.levels - skip
        bottom = self.decode_fea(dec, n_bottom * 3, with_lo=False)
        coords = np.concatenate((np.zeros((n_bottom, 1), np.int64), bottom.reshape(-1, 3)), 1)
        cur = Sp(np.ones((n_bottom, 1), np.float32), coords, 1 << levels)
        for idx in range(levels, 0, -1):
            pre, n_lat, lossless = self._block(idx, skip)
            cur = self.trunk(pre, cur)
            for j in range(n_lat):
                z = self.decode_fea(dec, len(cur.C) * cfg.compressed_channels).astype(np.float32).reshape(len(cur.C), -1)
                cur = self.absorb(f'{pre}.transforms.{j}', cur, z)
            logits = self.predict(pre, cur, lossless)
            if lossless:
                sym = np.empty(len(logits), np.uint16)
                dec.decode(quantize_pmf(logits, True), sym)
                mask = self._bits(sym)
            else:
                self.trace[f'dec_logits{idx}'] = logits
                mask = top_children(logits, points_num.pop())
            children = self._children(cur.C, mask)
            if idx == 1:
                return children[:, 1:] + offset
            cur = self.expand(pre, cur, mask, children, None)
