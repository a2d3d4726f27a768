"""Octree geometry codec with coded latents and top-k (lossy) finest levels: module tree, parameter names and bitstream of
/root/reference/models/convolutional/lossy_coord_v3/model.py (predictor :43-264, encoders :350-365, codec :547-681,
side information :513-545, residual block / sequences :684-710, latent prior :724-753).

Coarse to fine, one predictor per octree level.  A *lossless* level codes the 8-bit child occupancy of every voxel as a
255-ary symbol under a predicted distribution; levels with `num_latents` additionally receive up to two 1-channel latents
computed by the encoder network from the true geometry below, rounded and coded under their own histogram (sent as side
information).  The finest levels may be *lossy*: nothing is coded, the decoder keeps each voxel's best child plus the
globally best children up to the level's true point count (3 bytes in the header).

Float convolutions and linears run on fpcc_conv_f32 over the hash lookup tables of fastpcc_amd.int_sparse_conv, exactly as
in the float twin of the LiDAR codec (codecs/lossl_coord): the decoder recomputes the encoder's activations bit for bit
because every output element is one fixed-order FMA chain.  Softmax -> 16-bit CDF rows are tensor arithmetic as in the
reference (:501-509) and go to the host coder whole; rANS is the reference's single LIFO stream (rans_coder.py)."""
import io
import math
import time
from typing import List, Optional, Tuple

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import hipops as ops
from ...data import PCData
from ...entropy_models import _NoisyDeepFactorized, make_parameters
from ...int_sparse_conv import Conv3d, SparseTensor
from ...rans_coder import RansDecoder, RansEncoder
from ..lossl_coord.model import Block, SparseSequential
from ..lossl_coord_int import model as int_model
from ..lossl_coord_int.model import _as_occ, _bits_of, _children_of, _symbols_of
from .model_config import Config

log2_e = math.log2(math.e)
LATENT_BOUND = 20.0          # latents are clipped to [-20, 20] in training (:28-40): 41 values at most reach the coder


class EntropyModel(nn.Module):
    """Noisy deep-factorised prior of one latent (:724-753).  Only training reads it -- at test time the latent's histogram
    is the coding distribution -- but its parameters are part of the state_dict."""

    def __init__(self, batch_shape: torch.Size = torch.Size([1]), init_scale: float = 10,
                 num_filters: Tuple[int, ...] = (1, 3, 3, 3, 3, 1)):
        super().__init__()
        self.batch_shape = batch_shape
        self.prior_weights, self.prior_biases, self.prior_factors = make_parameters(batch_shape.numel(), init_scale, num_filters)

    def forward(self, x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """x + U(-.5, .5) and its cost in bits per element under the prior"""
        x = x + torch.empty_like(x).uniform_(-0.5, 0.5)
        prior = _NoisyDeepFactorized(self.batch_shape, self.prior_weights, self.prior_biases, self.prior_factors, 1.0)
        return x, prior.log_prob(x) * -log2_e


class _Bound(torch.autograd.Function):
    """clip to [-bound, bound]; outside the interval the gradient is replaced by +-1 so that the value is pulled back
    whatever the loss says (:28-40)"""

    @staticmethod
    def forward(ctx, x, bound):
        ctx.save_for_backward(x, bound)
        return torch.clip(x, -bound, bound)

    @staticmethod
    def backward(ctx, g):
        x, bound = ctx.saved_tensors
        g = torch.where(x > bound, torch.ones_like(g), g)
        g = torch.where(x < -bound, -torch.ones_like(g), g)
        return g, None


def top_children(logits: torch.Tensor, points_num: int) -> torch.Tensor:
    """[n, 8] child logits -> bool mask: every voxel's best child (ties included) and every child whose logit exceeds the
    (8n - points_num)-th smallest of the level (:216-221)"""
    mask = logits == logits.amax(1, keepdim=True)
    k = logits.shape[0] * 8 - int(points_num)
    if k >= 1:
        mask |= logits > torch.kthvalue(logits.reshape(-1), k).values
    else:
        mask |= True
    return mask


class OneScalePredictor(nn.Module):
    def __init__(self, channels, num_latents=0, if_pred_oct_lossl=True, if_upsample=True, allow_single_ch=False,
                 coord_recon_loss_factor=None, compressed_channels=None):
        super().__init__()
        self.compressed_channels = compressed_channels
        if allow_single_ch:
            self.dec_init = Conv3d(1, channels, 3, 1, 1, bias=True)
        self.dec = Block(channels)
        self.transforms = nn.ModuleList()
        self.num_latents = num_latents
        for _ in range(num_latents):
            self.transforms.append(nn.ModuleList((
                SparseSequential(nn.Linear(channels, channels, bias=True), nn.PReLU()),
                SparseSequential(nn.Linear(channels * 2, channels, bias=True), nn.PReLU(),
                                 Conv3d(channels, channels, 3, 1, 1, bias=True), nn.PReLU(),
                                 Conv3d(channels, compressed_channels, 3, 1, 1, bias=True)),
                SparseSequential(nn.Linear(compressed_channels, channels, bias=True), nn.PReLU()),
                SparseSequential(nn.Linear(channels * 2, channels, bias=True), nn.PReLU(), Block(channels)),
                EntropyModel(batch_shape=torch.Size([compressed_channels]), init_scale=10))))
        self.if_pred_oct_lossl = if_pred_oct_lossl
        self.if_upsample = if_upsample                     # False only for stride 2 -> 1
        self.coord_recon_loss_factor = coord_recon_loss_factor
        head = nn.Linear(channels, 255, bias=True) if if_pred_oct_lossl else Conv3d(channels, 8, 3, 1, 1, bias=True)
        self.pred = SparseSequential(Conv3d(channels, channels, 3, 1, 1, bias=True), nn.PReLU(), head)
        self.upsample = SparseSequential(nn.Linear(channels + 8, channels, bias=True), nn.PReLU(), Block(channels),
                                         nn.Linear(channels, channels * 8, bias=True)) if if_upsample else None

    # -- shared pieces ---------------------------------------------------------------------------------------------
    def _trunk(self, cur_rec: SparseTensor) -> SparseTensor:
        if cur_rec.F.shape[1] == 1:
            cur_rec = self.dec_init(cur_rec)
        return self.dec(cur_rec)

    @staticmethod
    def _absorb(cur_rec: SparseTensor, latent: SparseTensor, widen, dec) -> SparseTensor:
        """reconstruction features <- block(cat(features, widen(latent)))"""
        cur_rec.F = torch.cat((cur_rec.F, widen(latent).F), 1)
        return dec(cur_rec)

    def _expand(self, cur_rec: SparseTensor, bits: torch.Tensor, child_coords: torch.Tensor) -> SparseTensor:
        """features of the children selected by `bits` ([n, 8] bool): cat(features, bits) -> upsample -> [n, 8, C] -> rows"""
        occ = _as_occ(bits, child_coords.shape[0])              # bits, (row, octant) pairs: one kernel (fpcc_octree_children)
        cur_rec.F = torch.cat((cur_rec.F, occ.bits.to(torch.float32)), 1)
        f = self.upsample(cur_rec).F
        feats = f.reshape(f.shape[0], 8, f.shape[1] // 8)[occ.parent_row.long(), occ.octant.long()]
        return SparseTensor(feats, child_coords, tuple(s // 2 for s in cur_rec.stride))

    # -- training path (:90-169) -----------------------------------------------------------------------------------
    def forward(self, cur_rec: SparseTensor, cur_ref: SparseTensor, up_ref: SparseTensor, cur_bin: torch.Tensor,
                points_num: List[int], bin2oct_kernel, unfold_kernel, warmup: bool):
        """-> (features of the next level | None, latent rate terms, geometry loss of this level), all in bits per input
        point averaged over the batch.  cur_bin [n, 8]: true child occupancy of every voxel of cur_rec (zero rows for
        voxels a lossy level above kept in excess)."""
        device = cur_rec.F.device
        batch_size = len(points_num)
        batch_of = cur_rec.C[:, 0].long()
        per_row_points = torch.tensor(points_num, dtype=torch.float32, device=device)[batch_of]
        cur_rec = self._trunk(cur_rec)
        fea_losses = []
        for to_latent_a, to_latent_b, widen, dec, prior in self.transforms:
            ref = to_latent_a(cur_ref)
            ref.F = torch.cat((ref.F, cur_rec.F), 1)
            ref = to_latent_b(ref)
            noisy, bits = prior(_Bound.apply(ref.F, torch.tensor(LATENT_BOUND, device=device)))
            fea_losses.append((bits / per_row_points[:, None]).sum() * ((0.01 if warmup else 1.0) / batch_size))
            ref.F = noisy
            cur_rec = self._absorb(cur_rec, ref, widen, dec)
        cur_pred = self.pred(cur_rec).F
        if self.if_pred_oct_lossl:
            symbols = _symbols_of(cur_bin, bin2oct_kernel).long()
            geo_loss = (F.cross_entropy(cur_pred, symbols, reduction='none') / per_row_points).sum() * (log2_e / batch_size)
            kept, children = cur_bin.bool(), up_ref.C
        else:
            # the normaliser of a lossy level is the point count of the level it reconstructs (:124-133)
            if up_ref.stride[0] == 1:
                per_row_up, up_points = per_row_points, points_num
            else:
                up_points = torch.bincount(up_ref.C[:, 0].long(), minlength=batch_size).tolist()
                per_row_up = torch.tensor(up_points, dtype=torch.float32, device=device)[batch_of]
            geo_loss = (F.binary_cross_entropy_with_logits(cur_pred, cur_bin, reduction='none').sum(1) / per_row_up).sum() \
                * (self.coord_recon_loss_factor * log2_e / batch_size)
            kept = children = None
            if self.if_upsample:
                with torch.no_grad():
                    edges = torch.searchsorted(cur_rec.C[:, 0].contiguous(),
                                               torch.arange(batch_size + 1, device=device, dtype=cur_rec.C.dtype)).tolist()
                    kept = torch.cat([top_children(cur_pred[a:b], n) for a, b, n in zip(edges[:-1], edges[1:], up_points)], 0)
                    kept |= cur_bin.bool()                                  # the true children always survive in training
                    children = _children_of(cur_rec.C, unfold_kernel, kept)
        if not self.if_upsample:
            return None, fea_losses, geo_loss
        nxt = self._expand(cur_rec, kept, children)
        if self.if_pred_oct_lossl:
            nxt._caches = cur_ref._caches            # same coordinates as the encoder's level: its kernel maps are reused
        return nxt, fea_losses, geo_loss

    # -- test-time paths -------------------------------------------------------------------------------------------
    def compress(self, cur_rec: SparseTensor, cur_ref: SparseTensor, up_ref: SparseTensor, cur_bin: torch.Tensor,
                 bin2oct_kernel, if_upsample):
        """-> (features of the next level | None, rounded latents, 255-ary logits | None, symbols | None); all None from the
        first lossy level on: the encoder's work ends there (:171-202)"""
        if not self.if_pred_oct_lossl:
            if len(self.transforms):
                raise NotImplementedError('latents on a lossy level are never written by the reference encoder (:590-592) '
                                          'while its decoder reads them: such a configuration cannot be decoded')
            return None, [], None, None
        cur_rec = self._trunk(cur_rec)
        rounded = []
        for to_latent_a, to_latent_b, widen, dec, _ in self.transforms:
            ref = to_latent_a(cur_ref)
            ref.F = torch.cat((ref.F, cur_rec.F), 1)
            ref = to_latent_b(ref)
            ref.F = ref.F.round()
            rounded.append(ref.F)
            cur_rec = self._absorb(cur_rec, ref, widen, dec)
        cur_pred = self.pred(cur_rec).F
        cur_oct = _symbols_of(cur_bin, bin2oct_kernel)
        if if_upsample:
            cur_rec = self._expand(cur_rec, cur_bin.bool(), up_ref.C)
            cur_rec._caches = cur_ref._caches
        return cur_rec, rounded, cur_pred, cur_oct

    def decompress(self, cur_rec: SparseTensor, cached_points_num: List[int], bin2oct_kernel, unfold_kernel,
                   rans_decode_fea, rans_decode_oct, if_upsample):
        """-> features of the next level, or the reconstructed [m, 3] coordinates after the last one (:204-245)"""
        cur_rec = self._trunk(cur_rec)
        for _, _, widen, dec, _ in self.transforms:
            latent = SparseTensor(rans_decode_fea(cur_rec.C.shape[0] * self.compressed_channels, cur_rec.F.device)
                                  .reshape(cur_rec.C.shape[0], -1), cur_rec.C, cur_rec.stride, cur_rec.spatial_range)
            latent._caches = cur_rec._caches
            cur_rec = self._absorb(cur_rec, latent, widen, dec)
        cur_pred = self.pred(cur_rec).F
        if self.if_pred_oct_lossl:
            cur_bin = _bits_of(rans_decode_oct(cur_pred), bin2oct_kernel)
        else:
            cur_bin = top_children(cur_pred, cached_points_num.pop())
        children = _children_of(cur_rec.C, unfold_kernel, cur_bin)
        if if_upsample:
            return self._expand(cur_rec, cur_bin, children)
        return children[:, 1:]


class Model(nn.Module):
    @staticmethod
    def params_divider(s: str) -> int:
        return 1 if '.prior_weights.' in s or '.prior_biases.' in s or '.prior_factors.' in s else 0

    def __init__(self, cfg: Config):
        super().__init__()
        cfg.check()
        self.cfg = cfg
        self.max_downsample_times = int(np.log2(cfg.max_stride))
        ch = cfg.channels
        self.blocks_enc = nn.ModuleList()
        for idx in range(len(cfg.num_latents)):
            if all(v == 0 for v in cfg.num_latents[idx:]):
                break
            if idx == 0:
                block = Fold()
            elif idx == 1:
                block = SparseSequential(Conv3d(8, ch, 3, 1, bias=True), nn.PReLU(), Conv3d(ch, ch, 2, 2, bias=True), Block(ch))
                if ch >= 256:
                    block.insert(3, nn.PReLU())
            else:
                block = SparseSequential(Conv3d(ch, ch, 2, 2, bias=True), Block(ch))
            self.blocks_enc.append(block)
        self.block_dec_recurrent = OneScalePredictor(ch, 0, True, True, True)
        self.blocks_dec = nn.ModuleList()
        for idx, (n_lat, lossl) in enumerate(zip(cfg.num_latents, cfg.lossl_geo_upsample)):
            self.blocks_dec.append(OneScalePredictor(ch, n_lat, bool(lossl), idx != 0,
                                                     coord_recon_loss_factor=cfg.coord_recon_loss_factor,
                                                     compressed_channels=cfg.compressed_channels))
        fold = torch.zeros(8, 8, 1, dtype=torch.int8)
        fold.reshape(8, 8)[...] = torch.eye(8, dtype=torch.int8)
        self.register_buffer('fold2bin_kernel', fold, persistent=False)
        self.register_buffer('bin2oct_kernel', torch.arange(7, -1, -1, dtype=torch.int32), persistent=False)
        self.register_buffer('unfold_kernel', torch.tensor(
            [(0, dx, dy, dz) for dx in (0, 1) for dy in (0, 1) for dz in (0, 1)], dtype=torch.int32)[None], persistent=False)
        # fixed near-uniform CDFs of the side information (:383-388)
        self.fea_side_info_cdf1 = np.arange(2, 65537, dtype=np.int64).astype(np.uint16)[None].copy()
        self.fea_side_info_cdf2 = (np.arange(1, 129, dtype=np.int64) * 512).astype(np.uint16)[None].copy()
        self.fea_side_info_cdf1[:, -1] = 65535
        self.fea_side_info_cdf2[:, -1] = 65535
        self.rans_encoder = RansEncoder(32 * 1024 * 1024)
        self.rans_decoder = RansDecoder()
        self.trace: Optional[dict] = None          # tests set a dict: per level logits / latents / symbols of compress()

    # ---------------------------------------------------------------------------------------------------------------
    def forward(self, pc_data: PCData):
        if self.training:
            return self.train_forward(pc_data.xyz, pc_data.points_num, getattr(pc_data, 'training_step', 0))
        if pc_data.batch_size != 1:
            raise ValueError('Only supports batch size == 1 during testing.')
        return self.test_forward(pc_data)

    def _true_bits_on(self, coarse: SparseTensor, fine: SparseTensor) -> torch.Tensor:
        """[n, 8] float: which children of every voxel of `coarse` exist in `fine` (the reference re-runs its fold
        convolution onto substituted output coordinates for this, :437-441)"""
        from ...int_sparse_conv import _kernel_table
        with torch.no_grad():
            _, table = _kernel_table(fine.C, coarse.C, (2, 2, 2), (2, 2, 2), None)
            return (table[:coarse.C.shape[0]] > 0).to(torch.float32)

    def train_forward(self, xyz: torch.Tensor, points_num: List[int], training_step: int) -> dict:
        """rate of the lossless levels + latent rates + weighted occupancy cross-entropy of the lossy levels, in bits per
        input point (:411-455).  xyz int32 [N, 4], per sample Morton ('zyx') sorted and unique, samples in batch order."""
        warmup = training_step < self.cfg.warmup_steps
        org = self.get_init_pc(xyz.contiguous(), 1)
        levels = self.max_downsample_times
        bins = [org]
        for _ in range(levels):
            bins.append(self.get_bin(bins[-1]))
        strided = [org, bins[1]] if len(self.blocks_enc) else [org]
        for block in self.blocks_enc[1:]:
            strided.append(block(strided[-1]))
        strided += bins[len(strided):]
        top = strided[-1]
        cur_rec = SparseTensor(org.F[:top.C.shape[0]], top.C, (2 ** levels,) * 3)
        cur_rec._caches = org._caches
        exact = True                                    # cur_rec still sits on the true coordinates of its level
        losses = {}
        for idx in range(levels, 0, -1):
            block = self._block(idx, self.blocks_dec)
            cur_bin = bins[idx].F if exact else self._true_bits_on(cur_rec, strided[idx - 1])
            cur_rec, fea_losses, geo_loss = block(cur_rec, strided[idx], strided[idx - 1], cur_bin, points_num,
                                                  self.bin2oct_kernel, self.unfold_kernel, warmup)
            exact = exact and block.if_pred_oct_lossl
            losses[f'stride{2 ** idx}_geo_loss'] = geo_loss
            for i, v in enumerate(fea_losses):
                losses[f'stride{2 ** idx}_fea{i}_loss'] = v
        total = sum(losses.values())
        out = {k: v.item() for k, v in losses.items()}
        out['loss'] = total
        return out

    @staticmethod
    def get_init_pc(xyz: torch.Tensor, stride: int = 1) -> SparseTensor:
        # coordinates are Morton ('zyx') sorted and unique
        return SparseTensor(torch.ones((xyz.shape[0], 1), dtype=torch.float32, device=xyz.device), xyz, (stride,) * 3)

    @torch.no_grad()
    def get_bin(self, sp: SparseTensor, ones_feats: Optional[torch.Tensor] = None) -> SparseTensor:
        """next octree level: parent coordinates and the 8 child-occupancy bits per parent as 0.0 / 1.0 (:390-409); the bits
        come from the integer fold kernel (exact)"""
        ones8 = torch.ones((sp.C.shape[0], 1), dtype=torch.int8, device=sp.C.device)
        ret = int_model.Model.get_bin(self, sp, ones8)
        ret.F = ret.F.to(torch.float32)
        return ret

    # -- entropy coding (:501-545) -------------------------------------------------------------------------------------
    @staticmethod
    def batch_quantize_pmf_torch(pmfs: torch.Tensor, softmax: bool = True) -> torch.Tensor:
        """[n, c] logits (or probabilities) -> int32 CDF rows without the leading zero, every frequency >= 1"""
        if softmax:
            pmfs = F.softmax(pmfs.float(), dim=-1)
        pmfs = pmfs.mul(65536 - pmfs.shape[1]).floor_().add_(1)
        pmfs.cumsum_(-1)
        pmfs[:, -1] = 65535
        return pmfs.to(torch.int32)

    @staticmethod
    def _to_host_u16(t: torch.Tensor) -> np.ndarray:
        h = torch.empty(t.shape, dtype=torch.int32, pin_memory=t.is_cuda)
        h.copy_(t, non_blocking=True)
        if t.is_cuda:
            torch.cuda.current_stream().synchronize()
        return h.numpy().astype(np.uint16)

    def histogram_cdf(self, values: torch.Tensor) -> np.ndarray:
        """coding distribution of a non-negative integer array: its own histogram (:566-567, :603-604)"""
        pmf = torch.bincount(values.to(torch.int32), minlength=2) / values.numel()
        return self._to_host_u16(self.batch_quantize_pmf_torch(pmf[None], False)[0])

    def rans_encode_oct(self, quantized_cdfs: np.ndarray, values: np.ndarray) -> int:
        return self.rans_encoder.encode(quantized_cdfs, values)

    def rans_decode_oct(self, logits: torch.Tensor) -> torch.Tensor:
        rows = self._to_host_u16(self.batch_quantize_pmf_torch(logits))
        out = np.empty(rows.shape[0], dtype=np.uint16)
        self.rans_decoder.decode(rows, out)
        symbols = torch.from_numpy(out.astype(np.int16)).to(logits.device)
        symbols._fpcc_children = int_model._children_count(out)
        return symbols

    def rans_encode_fea(self, quantized_cdf: np.ndarray, rounded: np.ndarray, rounded_min: Optional[int] = None):
        """symbols under one shared CDF, then the CDF itself (entries - 1 under a flat 16-bit model), its length and the
        value offset (flat 7-bit model): the decoder pops them in the opposite order"""
        self.rans_encoder.encode(quantized_cdf[None], rounded)
        self.rans_encoder.encode(self.fea_side_info_cdf1, quantized_cdf[:-1] - 1)
        if len(quantized_cdf) - 2 > self.fea_side_info_cdf2.shape[1] - 1:
            raise ValueError(f'alphabet of {len(quantized_cdf)} values does not fit the side information')
        self.rans_encoder.encode(self.fea_side_info_cdf2, np.array((len(quantized_cdf) - 2,), dtype=np.uint16))
        if rounded_min is not None:
            if not 0 <= rounded_min < self.fea_side_info_cdf2.shape[1]:
                raise ValueError(f'latent offset {rounded_min} does not fit the side information')
            self.rans_encoder.encode(self.fea_side_info_cdf2, np.array((rounded_min,), dtype=np.uint16))

    def rans_decode_fea(self, length: int, device=None, decode_rounded_min: bool = True):
        """-> float tensor on `device` (latents, offset removed) or, with decode_rounded_min False, the raw uint16 array"""
        rounded_min = np.zeros(1, dtype=np.uint16)
        if decode_rounded_min:
            self.rans_decoder.decode(self.fea_side_info_cdf2, rounded_min)
        cdf_len = np.empty(1, dtype=np.uint16)
        self.rans_decoder.decode(self.fea_side_info_cdf2, cdf_len)
        cdf = np.empty(int(cdf_len[0]) + 1, dtype=np.uint16)
        self.rans_decoder.decode(self.fea_side_info_cdf1, cdf)
        cdf = np.pad(cdf + 1, (0, 1))
        cdf[-1] = 65535
        decoded = np.empty(length, dtype=np.uint16)
        self.rans_decoder.decode(cdf[None], decoded)
        if not decode_rounded_min:
            return decoded
        return torch.from_numpy(decoded.astype(np.float32) - np.float32(rounded_min[0])).to(device)

    # -- codec (:547-681) ------------------------------------------------------------------------------------------------
    def _block(self, idx: int, blocks) -> OneScalePredictor:
        return self.block_dec_recurrent if idx > len(blocks) else blocks[idx - 1]

    @torch.no_grad()
    def compress(self, xyz: torch.Tensor) -> bytes:
        if not xyz.is_cuda:
            raise RuntimeError('compress() runs on the GPU; move the coordinates there first')
        coord_offset = xyz[:, 1:].amin(0)
        xyz = xyz - F.pad(coord_offset, (1, 0))
        _, perm = ops.sort_keys(ops.morton3d_encode(xyz[:, 1:], (2, 1, 0)))       # 'zyx': z on Morton bit 0
        xyz = xyz[perm.long()].contiguous()
        org = self.get_init_pc(xyz, 1)
        skip = self.cfg.skip_top_scales_num
        blocks_enc, blocks_dec = self.blocks_enc[skip:], self.blocks_dec[skip:]
        levels = self.max_downsample_times - skip
        if skip and len(self.blocks_enc):
            raise NotImplementedError('skip_top_scales_num with latent levels: the encoder stack starts at the fold of stride 1')

        # octree occupancy of every level first (this also fixes the coordinates the strided encoder convolutions land on),
        # then the encoder features of the levels that code latents
        bins = [org]
        for _ in range(levels):
            bins.append(self.get_bin(bins[-1]))
        strided = [org, bins[1]] if len(blocks_enc) else [org]
        for block in blocks_enc[1:]:
            strided.append(block(strided[-1]))
        strided += bins[len(strided):]

        lossy_levels = next((i for i, v in enumerate(self.cfg.lossl_geo_upsample) if v == 1), len(self.cfg.lossl_geo_upsample))
        cached_points_num = [strided[i].C.shape[0] for i in range(lossy_levels)]
        top = strided[-1]
        bottom = top.C[:, 1:].reshape(-1)
        cur_rec = SparseTensor(org.F[:top.C.shape[0]], top.C, (2 ** levels,) * 3)
        cur_rec._caches = org._caches

        pending = []
        for idx in range(levels, 0, -1):
            block = self._block(idx, blocks_dec)
            cur_rec, rounded, logits, symbols = block.compress(cur_rec, strided[idx], strided[idx - 1], bins[idx].F,
                                                               self.bin2oct_kernel, if_upsample=idx != 1)
            if logits is None:
                break
            if self.trace is not None:
                self.trace[f'logits{idx}'], self.trace[f'symbols{idx}'] = logits, symbols
                for j, f in enumerate(rounded):
                    self.trace[f'latent{idx}.{j}'] = f.clone()
            latents = []
            for f in rounded:
                lo = int(-f.min().item())
                shifted = (f + lo).reshape(-1)
                latents.append((self.histogram_cdf(shifted), self._to_host_u16(shifted.to(torch.int32)), lo))
            pending.append((latents, self._to_host_u16(self.batch_quantize_pmf_torch(logits)),
                            self._to_host_u16(symbols.to(torch.int32))))
        while pending:                                             # finest level first: the decoder pops coarse -> fine
            latents, rows, symbols = pending.pop()
            self.rans_encode_oct(rows, symbols)
            while latents:
                self.rans_encode_fea(*latents.pop())
        self.rans_encode_fea(self.histogram_cdf(bottom), self._to_host_u16(bottom.to(torch.int32)))

        with io.BytesIO() as bs:
            for v in coord_offset.tolist():
                bs.write(int(v).to_bytes(2, 'little'))
            bs.write((bottom.shape[0] // 3).to_bytes(2, 'little'))
            for n in cached_points_num:
                bs.write(int(n).to_bytes(3, 'little'))
            bs.write(self.rans_encoder.flush())
            return bs.getvalue()

    def compress_partitions(self, batched_coord: List[torch.Tensor]) -> bytes:
        parts = [self.compress(p) for p in batched_coord[1:]]            # [0] holds the unpartitioned cloud
        return b''.join(len(s).to_bytes(3, 'little') + s for s in parts)

    @torch.no_grad()
    def decompress(self, compressed_bytes: bytes) -> torch.Tensor:
        device = self.fold2bin_kernel.device
        coord_offset = [int.from_bytes(compressed_bytes[i:i + 2], 'little') for i in (0, 2, 4)]
        n_bottom = int.from_bytes(compressed_bytes[6:8], 'little')
        pos = 8
        cached_points_num = []
        for v in self.cfg.lossl_geo_upsample:
            if v == 1:
                break
            cached_points_num.append(int.from_bytes(compressed_bytes[pos:pos + 3], 'little'))
            pos += 3
        self.rans_decoder.flush(compressed_bytes[pos:])
        skip = self.cfg.skip_top_scales_num
        blocks_dec = self.blocks_dec[skip:]
        levels = self.max_downsample_times - skip

        bottom = torch.from_numpy(self.rans_decode_fea(n_bottom * 3, decode_rounded_min=False).astype(np.int32)).reshape(-1, 3)
        cur_rec = self.get_init_pc(F.pad(bottom, (1, 0, 0, 0)).to(device), 2 ** levels)
        for idx in range(levels, 0, -1):
            cur_rec = self._block(idx, blocks_dec).decompress(
                cur_rec, cached_points_num, self.bin2oct_kernel, self.unfold_kernel, self.rans_decode_fea,
                self.rans_decode_oct, if_upsample=idx != 1)
        return cur_rec + torch.tensor(coord_offset, device=device, dtype=torch.int32)[None]

    def decompress_partitions(self, concat_bytes: bytes) -> torch.Tensor:
        out, pos = [], 0
        while pos != len(concat_bytes):
            length = int.from_bytes(concat_bytes[pos:pos + 3], 'little')
            out.append(self.decompress(concat_bytes[pos + 3: pos + 3 + length]))
            pos += 3 + length
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
        return {'compressed_bytes': data, 'pred': recon, 'encode time': t1 - t0, 'decode time': t2 - t1}


class Fold(nn.Module):
    """stride-2 identity-kernel convolution: 8 child-occupancy channels per parent (:248-267).  Parameter-free; the codec
    takes the same bits from Model.get_bin, so this module exists for the module tree (blocks_enc.0) and for callers that
    run the encoder stack themselves."""

    def forward(self, sp: SparseTensor) -> SparseTensor:
        out_coords = sp.C.clone()
        out_coords[:, 1:] >>= 1
        out_coords = torch.unique_consecutive(out_coords, dim=0)
        fold = torch.zeros(8, 8, 1, dtype=torch.int8, device=sp.C.device)
        fold.reshape(8, 8)[...] = torch.eye(8, dtype=torch.int8, device=sp.C.device)
        holder = type('_Holder', (), {'fold2bin_kernel': fold})()
        ret = int_model.Model.get_bin(holder, sp, torch.ones((sp.C.shape[0], 1), dtype=torch.int8, device=sp.C.device))
        ret.F = ret.F.to(torch.float32)
        return ret
