"""Hierarchical lossless coder of a sparse latent: occupancy of every finer level (binary rANS under predicted
probabilities) + one-channel integer residual features (rANS under their empirical histogram).

Interface and bitstream of `GeoLosslessEntropyModel`
(/root/reference/models/convolutional/lossy_coord_lossy_color/geo_lossl_em.py:20-317; layout in SURVEY.md appendix B).
What differs is where the work happens:
  * encode: nothing leaves the GPU inside the level loop.  Occupancy masks, 16-bit probabilities and residual symbols
    of all levels are left in device buffers and fetched with ONE synchronising copy after the last level (the
    reference does ~18 blocking .cpu() calls); the six occupancy streams are then coded concurrently on host threads;
  * decode: one device->host (probabilities) and one host->device (mask) transfer per occupancy level -- the
    dependency chain of the format;
  * coordinate membership (`get_coord_mask`) is a read of the pyramid's child_row table, not a kernel-map query.
"""
import io
import math
from typing import List, Optional, Tuple

import numpy as np
import torch
import torch.nn as nn

from .. import engine as ME
from .. import hipops as ops
from .._native import host, host_check
from ..bitstream import BytesListUtils, bytes_to_int, int_to_bytes
from ..entropy_models import NoisyDeepFactorizedEntropyModel
from ..rans_coder import BinaryRansCoder, IndexedRansCoder


class GeoLosslessEntropyModel(nn.Module):
    def __init__(self, compressed_channels: int, bottleneck_process: str, bottleneck_scaler: int,
                 skip_encoding_fea: int, encoder: nn.Module, residual_block: nn.Module, decoder_block: nn.Module,
                 hyper_decoder_coord: nn.Module, hyper_decoder_fea: nn.Module):
        super().__init__()
        if compressed_channels != 1:
            raise NotImplementedError('one coded channel per level, as in every in-scope configuration')
        self.compressed_channels = compressed_channels
        self.broadcast_shape_bytes = 3
        self.bottleneck_scaler = bottleneck_scaler
        self.skip_encoding_fea = skip_encoding_fea
        # The noisy deep-factorised bottleneck prices the residual features in the rate objective (forward()); the coded
        # path uses empirical histograms instead.  Same sub-module name / state-dict keys as the reference (:38-48).
        self.bottom_fea_entropy_model = NoisyDeepFactorizedEntropyModel(
            batch_shape=torch.Size([compressed_channels]), coding_ndim=2, bottleneck_process=bottleneck_process,
            bottleneck_scaler=bottleneck_scaler, init_scale=10, broadcast_shape_bytes=(self.broadcast_shape_bytes,))
        self.rans_coder = IndexedRansCoder(False, 1)
        self.binary_rans_coder = BinaryRansCoder(1)
        assert len(encoder) == len(residual_block) == len(decoder_block) == len(hyper_decoder_fea)
        self.encoder = encoder
        self.residual_block = residual_block
        self.decoder_block = decoder_block
        self.hyper_decoder_coord = hyper_decoder_coord
        self.hyper_decoder_fea = hyper_decoder_fea
        self.host_threads = 8
        self.keep_symbols = False
        self.last_symbols = None

    # -- rANS of an integer array under its own histogram (geo_lossl_em.py:59-93) -------------------------------------
    def rans_encode_with_cdf(self, target: np.ndarray, bs: io.BytesIO, offset: Optional[int] = None):
        bs.write(int_to_bytes(int(target.shape[0]), self.broadcast_shape_bytes))
        if offset is None:
            offset = int(target.min())
            bs.write(int_to_bytes(-offset, 1))
        hist = np.bincount((target - offset).reshape(-1))
        self.rans_coder.init_with_pmfs(hist[None].astype(np.float64), np.array([offset], dtype=np.int32))
        cdf = self.rans_coder.get_cdfs()[0]
        bs.write(int_to_bytes(len(cdf) - 2, 1))
        for edge in cdf[1:-1]:
            bs.write(int_to_bytes(edge, 2))
        payload = self.rans_coder.encode(target.reshape(1, -1))[0]
        bs.write(int_to_bytes(len(payload), 3))
        bs.write(payload)

    def rans_decode_with_cdf(self, bs: io.BytesIO, offset: Optional[int] = None, channels: Optional[int] = None) \
            -> Tuple[np.ndarray, List[int]]:
        rows = bytes_to_int(bs.read(self.broadcast_shape_bytes))
        if offset is None:
            offset = -bytes_to_int(bs.read(1))
        inner = bytes_to_int(bs.read(1))
        cdf = [0, *(bytes_to_int(bs.read(2)) for _ in range(inner)), 1 << 16]
        self.rans_coder.init_with_quantized_cdfs([cdf], np.array([offset], dtype=np.int32))
        payload = bs.read(bytes_to_int(bs.read(3)))
        width = channels or self.compressed_channels
        out = np.empty((1, rows * width), np.int32)
        self.rans_coder.decode([payload], out)
        return out.reshape(rows, width), cdf

    # -- rate objective (geo_lossl_em.py:115-158) -----------------------------------------------------------------------
    def forward(self, y_top: ME.SparseTensor, batch_size: int = 1):
        """Training-mode forward: returns (reconstructed top features, {'fea_bottom_bits_loss', 'coord_i_bits_loss',
        'fea_i_bits_loss'}).  The sparse convolutions of this build are forward-only, so the terms carry gradients
        only w.r.t. the entropy-model parameters; back-propagation through the convolutions is not built yet."""
        if not self.training:
            raise RuntimeError('forward() evaluates the training objective; use compress() / decompress() for coding')
        cm = y_top.coordinate_manager
        *feas, bottom = self.encoder(y_top, batch_size)
        loss = {}
        tilde, d = self.bottom_fea_entropy_model(bottom.F[None])
        loss['fea_bottom_bits_loss'] = d['bits_loss']
        lower = ME.SparseTensor(tilde[0], coordinate_map_key=bottom.coordinate_map_key, coordinate_manager=cm)
        for idx in range(len(feas) - 1, -1, -1):
            fea = feas[idx]
            target_key = fea.coordinate_map_key
            target_map = cm._map(target_key)
            if cm._map(lower.coordinate_map_key) is not target_map:
                logits = self.hyper_decoder_coord[idx](lower)
                mask = ops.child_mask(target_map.child_row).to(torch.float)
                loss[f'coord_{idx}_bits_loss'] = nn.functional.binary_cross_entropy_with_logits(
                    logits.F.view(-1), mask, reduction='sum') / math.log(2)
            fea_pred = self.hyper_decoder_fea[idx](lower, target_key)
            if idx > self.skip_encoding_fea:
                res = self.residual_block[idx](fea, fea_pred)
                res_tilde, d = self.bottom_fea_entropy_model(res.F[None])
                loss[f'fea_{idx}_bits_loss'] = d['bits_loss']
                lower = self.decoder_block[idx](res_tilde[0], fea_pred)
            else:
                lower = self.decoder_block[idx](fea_pred)
        return lower, loss

    # -- compress -------------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def compress(self, y_top: ME.SparseTensor, batch_size: int = 1) -> bytes:
        cm = y_top.coordinate_manager
        *feas, bottom = self.encoder(y_top, batch_size)
        del y_top
        scale = float(self.bottleneck_scaler)
        bottom_f = bottom.F
        residual_syms = [ops.quantize_symbols_(bottom_f, scale)]         # rounds bottom_f in place
        occupancy: List[Tuple[torch.Tensor, torch.Tensor]] = []          # (mask u8, prob u16) per coded level
        lower = bottom
        bottom_map = cm._map(bottom.coordinate_map_key)

        for idx in range(len(feas) - 1, -1, -1):
            fea = feas[idx]
            feas[idx] = None
            target_key = fea.coordinate_map_key
            target_map = cm._map(target_key)
            if cm._map(lower.coordinate_map_key) is not target_map:
                if target_map.parent is not cm._map(lower.coordinate_map_key):
                    raise RuntimeError('pyramid levels are not parent and child')
                logits = self.hyper_decoder_coord[idx](lower)            # on the 8 candidate children of every voxel
                occupancy.append((ops.child_mask(target_map.child_row), ops.logit_to_prob16(logits.F.view(-1))))
                del logits
            elif self.hyper_decoder_coord[idx] is not None:
                raise RuntimeError('an occupancy predictor exists for a level that does not upsample')

            fea_pred = self.hyper_decoder_fea[idx](lower, target_key)
            if idx > self.skip_encoding_fea:
                res = self.residual_block[idx](fea, fea_pred).F
                del fea
                residual_syms.append(ops.quantize_symbols_(res, scale))
                lower = self.decoder_block[idx](res, fea_pred)
            else:
                lower = self.decoder_block[idx](fea_pred)
            del fea_pred
        del lower

        # one synchronising transfer for everything the host coders need
        dev = bottom_f.device
        sym_all = torch.cat(residual_syms)
        bottom_xyz = (cm.get_coordinates(bottom.coordinate_map_key)[:, 1:] >> bottom_map.level).contiguous()
        if occupancy:
            mask_all = torch.cat([m for m, _ in occupancy])
            prob_all = torch.cat([p for _, p in occupancy])
        else:
            mask_all = torch.empty(0, dtype=torch.uint8, device=dev)
            prob_all = torch.empty(0, dtype=torch.int16, device=dev)
        sym_h = torch.empty(sym_all.shape, dtype=sym_all.dtype, pin_memory=True)
        xyz_h = torch.empty(bottom_xyz.shape, dtype=bottom_xyz.dtype, pin_memory=True)
        mask_h = torch.empty(mask_all.shape, dtype=mask_all.dtype, pin_memory=True)
        prob_h = torch.empty(prob_all.shape, dtype=prob_all.dtype, pin_memory=True)
        sym_h.copy_(sym_all, non_blocking=True)
        xyz_h.copy_(bottom_xyz, non_blocking=True)
        mask_h.copy_(mask_all, non_blocking=True)
        prob_h.copy_(prob_all, non_blocking=True)
        torch.cuda.current_stream().synchronize()

        sizes = [m.numel() for m, _ in occupancy]
        coord_bytes_list = self._encode_occupancy(mask_h.numpy(), prob_h.numpy().view(np.uint16), sizes)
        if self.keep_symbols:      # test hook: what went into the coders
            self.last_symbols = {'residual': sym_h.numpy().copy(), 'occupancy': mask_h.numpy().copy(),
                                 'prob': prob_h.numpy().view(np.uint16).copy(), 'sizes': sizes}

        with io.BytesIO() as bs:
            bs.write(int_to_bytes(bottom_map.level, 1))                                 # log2(bottom stride)
            bs.write(int_to_bytes(bottom_map.n, self.broadcast_shape_bytes))
            self.rans_encode_with_cdf(sym_h.numpy().reshape(-1, 1), bs)
            bs.write(int_to_bytes(len(coord_bytes_list), 1))
            BytesListUtils.concat_bytes_list(coord_bytes_list, bs)
            self.rans_encode_with_cdf(xyz_h.numpy(), bs, 0)
            return bs.getvalue()

    def _encode_occupancy(self, mask: np.ndarray, prob: np.ndarray, sizes: List[int]) -> List[bytes]:
        if not sizes:
            return []
        start = np.zeros(len(sizes) + 1, dtype=np.int64)
        np.cumsum(sizes, out=start[1:])
        cap = 4 * max(sizes) + 64
        out = np.empty((len(sizes), cap), dtype=np.uint8)
        lens = np.zeros(len(sizes), dtype=np.int64)
        host_check(host().fpcc_rans_binary_encode_multi(mask.ctypes.data, prob.ctypes.data, start.ctypes.data,
                                                        len(sizes), out.ctypes.data, cap, lens.ctypes.data,
                                                        self.host_threads))
        return [out[s, cap - int(lens[s]):].tobytes() for s in range(len(sizes))]

    # -- decompress -----------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def decompress(self, concat_bytes: bytes, cm: ME.CoordinateManager) -> ME.SparseTensor:
        dev = next(self.parameters()).device
        scale = float(self.bottleneck_scaler)
        with io.BytesIO(concat_bytes) as bs:
            bottom_level = bytes_to_int(bs.read(1))
            bottom_rows = bytes_to_int(bs.read(self.broadcast_shape_bytes))
            syms, _ = self.rans_decode_with_cdf(bs)
            n_streams = bytes_to_int(bs.read(1))
            coord_bytes_list = BytesListUtils.split_bytes_list(None, n_streams, bs) if n_streams else []
            bottom_xyz, _ = self.rans_decode_with_cdf(bs, 0, 3)

        res_all = torch.from_numpy(syms.astype(np.float32)).to(dev)
        if scale != 1.0:
            res_all /= scale
        coords = torch.zeros((bottom_rows, 4), dtype=torch.int32)
        coords[:, 1:] = torch.from_numpy(bottom_xyz) << bottom_level
        lower = ME.SparseTensor(res_all[:bottom_rows].contiguous(), coordinates=coords.to(dev),
                                tensor_stride=1 << bottom_level, coordinate_manager=cm)
        used = bottom_rows

        for idx in range(len(self.residual_block) - 1, -1, -1):
            occ_net = self.hyper_decoder_coord[idx]
            cur_map = cm._map(lower.coordinate_map_key)
            if occ_net is not None:
                logits = occ_net(lower)
                prob = ops.logit_to_prob16(logits.F.view(-1)).cpu().numpy().view(np.uint16)   # blocking D2H
                bits = np.empty((1, prob.size), dtype=np.bool_)
                self.binary_rans_coder.decode([coord_bytes_list.pop(0)], prob.reshape(1, -1), bits)
                mask = torch.from_numpy(bits.reshape(-1).view(np.uint8)).to(dev)      # H2D
                gen_id = logits.coordinate_map_key.get_key()[1]
                cur_map = cm._refine(cur_map, mask, gen_id + 'pruned')
                del logits
            target_key = cur_map.key
            fea_pred = self.hyper_decoder_fea[idx](lower, target_key)
            if idx > self.skip_encoding_fea:
                res = res_all[used: used + cur_map.n]
                used += cur_map.n
                lower = self.decoder_block[idx](res, fea_pred)
            else:
                lower = self.decoder_block[idx](fea_pred)
        if coord_bytes_list:
            raise ValueError('unused occupancy streams in the bitstream')
        if used != res_all.shape[0]:
            raise ValueError('residual symbols left over in the bitstream')
        return lower
