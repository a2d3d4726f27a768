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
import os
import math
import time
from typing import List, Optional, Tuple

import numpy as np
import torch
import torch.nn as nn

from .. import engine as ME
from .. import hipops as ops
from .._native import host, host_check
from ..bitstream import BytesListUtils, bytes_to_int, int_to_bytes
from ..coder_pool import CoderPool
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
        self.evaluate_unused_tail = False
        # compress(): feature chain first, occupancy predictors afterwards finest first (same bytes); FPCC_DEFER_OCCUPANCY=0: chain order
        self.defer_occupancy = os.environ.get('FPCC_DEFER_OCCUPANCY', '1') != '0'
        # experiment (FPCC_OCCUPANCY_STREAM=1, overrides defer_occupancy): every occupancy predictor is enqueued on a second stream the
        # moment its input exists and runs BESIDE the feature chain -- the predictors are leaves, nothing on the device reads their
        # result.  Same launches, same bytes (tests/test_gpu_codec_v2.py runs both); measured in profiles/r04/frames_in_flight.md
        self.occupancy_stream = os.environ.get('FPCC_OCCUPANCY_STREAM', '0') == '1'
        self._overlap = {}
        # True: occupancy levels are decoded by the device-side binary rANS decoder (fpcc_rans_binary_decode_dev) -- nothing
        # but the 4-byte count of occupied children leaves the GPU per level.  Same result; slower than the host path on the
        # large levels (profiles/r02/device_rans.md), hence off by default.
        self.device_decoder = False
        self.timing = None          # set to a dict to collect host-side wall-clock marks (seconds) of the last call

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
        'fea_i_bits_loss'}), differentiable end to end (convolutions through fastpcc_amd/autograd.py)."""
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
    def _overlap_state(self, dev: torch.device):
        st = self._overlap.get(dev)
        if st is None:
            st = {'pool': CoderPool(self.host_threads), 'side': torch.cuda.Stream(device=dev),
                  'flags': torch.zeros(64, dtype=torch.int32, pin_memory=True),
                  'ones': torch.ones(64, dtype=torch.int32, device=dev)}
            self._overlap[dev] = st
        return st

    @staticmethod
    def _ship(st, tensors: List[torch.Tensor], flag: Optional[int]) -> List[np.ndarray]:
        """stream-ordered device->pinned-host copies on the side stream, then (optionally) the flag that releases the
        host job reading them; the main stream is not blocked"""
        ev = torch.cuda.Event()
        ev.record()
        side = st['side']
        out = []
        with torch.cuda.stream(side):
            side.wait_event(ev)
            for t in tensors:
                h = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
                h.copy_(t, non_blocking=True)
                t.record_stream(side)
                out.append(h)
            if flag is not None:
                st['flags'][flag:flag + 1].copy_(st['ones'][flag:flag + 1], non_blocking=True)
        st['pinned'].extend(out)
        return [h.numpy() for h in out]

    def _read_coded(self, bs: io.BytesIO, offset: Optional[int]) -> Tuple[int, int, List[int], bytes]:
        rows = bytes_to_int(bs.read(self.broadcast_shape_bytes))
        if offset is None:
            offset = -bytes_to_int(bs.read(1))
        inner = bytes_to_int(bs.read(1))
        cdf = [0, *(bytes_to_int(bs.read(2)) for _ in range(inner)), 1 << 16]
        payload = bs.read(bytes_to_int(bs.read(3)))
        return rows, offset, cdf, payload

    def _write_coded(self, bs: io.BytesIO, rows: int, offset: int, cdf: List[int], payload: bytes, write_offset: bool):
        """byte layout of rans_encode_with_cdf (geo_lossl_em.py:59-74)"""
        bs.write(int_to_bytes(rows, self.broadcast_shape_bytes))
        if write_offset:
            bs.write(int_to_bytes(-offset, 1))
        bs.write(int_to_bytes(len(cdf) - 2, 1))
        for edge in cdf[1:-1]:
            bs.write(int_to_bytes(edge, 2))
        bs.write(int_to_bytes(len(payload), 3))
        bs.write(payload)

    @torch.no_grad()
    def compress(self, y_top: ME.SparseTensor, batch_size: int = 1) -> bytes:
        """Entropy coding runs on libfpcc_host's threads WHILE the GPU evaluates the following levels: every level's
        mask / probabilities (and, after the last residual level, all residual symbols) go to pinned memory on a side
        stream, followed by a flag that releases the host job (fastpcc_amd/coder_pool.py).  One wait at the end."""
        tm = self.timing
        if tm is not None:
            tm.clear()
            tm['enc_t0'] = time.perf_counter()
        cm = y_top.coordinate_manager
        *feas, bottom = self.encoder(y_top, batch_size)
        del y_top
        scale = float(self.bottleneck_scaler)
        bottom_f = bottom.F
        st = self._overlap_state(bottom_f.device)
        st['pinned'] = []
        st['flags'].zero_()
        pool: CoderPool = st['pool']
        flags = st['flags'].numpy().view(np.uint32)
        n_flags = 0
        residual_syms = [ops.quantize_symbols_(bottom_f, scale)]         # rounds bottom_f in place
        occupancy_h: List[Tuple[np.ndarray, np.ndarray]] = []            # (mask u8, prob u16) per coded level, host side
        lower = bottom
        bottom_map = cm._map(bottom.coordinate_map_key)
        bottom_xyz = (cm.get_coordinates(bottom.coordinate_map_key)[:, 1:] >> bottom_map.level).contiguous()
        (xyz_h,) = self._ship(st, [bottom_xyz], None)
        # Last level whose evaluation still feeds the bitstream: the reference keeps evaluating the feature predictors
        # below it and discards the result (`del lower_fea_recon`, geo_lossl_em.py:210); nothing reads them, so this
        # build stops there unless `evaluate_unused_tail` is set (identical bytes either way, tested).
        last_coded = min((i for i in range(len(feas)) if i > self.skip_encoding_fea or self.hyper_decoder_coord[i] is not None),
                         default=len(feas))
        last_residual = min((i for i in range(len(feas)) if i > self.skip_encoding_fea), default=len(feas))
        residual_job = None

        def ship_residuals():
            nonlocal n_flags
            (sym_h,) = self._ship(st, [torch.cat(residual_syms)], n_flags)
            job = pool.histogram_encode(sym_h, None, flags[n_flags:n_flags + 1])
            n_flags += 1
            return sym_h, job

        # The occupancy predictors are leaves of the top-down chain: level idx's predictor reads `lower` of that level and nothing
        # reads its result on the device.  Evaluated in chain order the largest one comes last, and the GPU idles while the host
        # pushes its symbols (709 K of the 1 M-voxel frame's, 1.6 ms) through the stream's single rANS state.  With `defer_occupancy`
        # the chain runs first and the predictors afterwards from the FINEST level to the coarsest: the long host passes start early
        # and run beside the remaining predictors.  Same launches, same inputs, same bytes; the streams are put back in level order.
        pending_occ: List[Tuple[int, ME.SparseTensor, object]] = []

        def code_occupancy(idx: int, lower_of_level: ME.SparseTensor, target_map) -> None:
            nonlocal n_flags
            logits = self.hyper_decoder_coord[idx](lower_of_level)       # on the 8 candidate children of every voxel
            mask_h, prob_h = self._ship(st, [ops.child_mask(target_map.child_row),
                                             ops.logit_to_prob16(logits.F.view(-1))], n_flags)
            prob_h = prob_h.view(np.uint16)
            pool.binary_encode(mask_h, prob_h, flags[n_flags:n_flags + 1])
            n_flags += 1
            occupancy_h.append((mask_h, prob_h))

        if last_residual == len(feas):
            residual_job = ship_residuals()
        for idx in range(len(feas) - 1, -1, -1):
            fea = feas[idx]
            feas[idx] = None
            target_key = fea.coordinate_map_key
            target_map = cm._map(target_key)
            if cm._map(lower.coordinate_map_key) is not target_map:
                if target_map.parent is not cm._map(lower.coordinate_map_key):
                    raise RuntimeError('pyramid levels are not parent and child')
                if self.occupancy_stream:
                    # what the predictor shares with the feature chain exists before the fork (built lazily, it would be built on
                    # whichever stream asks first): the 3x3x3 table of the level's map, parent of the generated map's table
                    cm._nbr27(cm._map(lower.coordinate_map_key))
                    fork = torch.cuda.Event()
                    fork.record()
                    occ = st.setdefault('occ', torch.cuda.Stream(device=bottom_f.device))
                    with torch.cuda.stream(occ):
                        occ.wait_event(fork)
                        code_occupancy(idx, lower, target_map)
                    lower.F.record_stream(occ)
                elif self.defer_occupancy:
                    pending_occ.append((idx, lower, target_map))
                else:
                    code_occupancy(idx, lower, target_map)
            elif self.hyper_decoder_coord[idx] is not None:
                raise RuntimeError('an occupancy predictor exists for a level that does not upsample')
            if idx <= last_coded and idx <= self.skip_encoding_fea and not self.evaluate_unused_tail:
                break

            fea_pred = self.hyper_decoder_fea[idx](lower, target_key)
            if idx > self.skip_encoding_fea:
                res = self.residual_block[idx](fea, fea_pred).F
                del fea
                residual_syms.append(ops.quantize_symbols_(res, scale))
                if idx == last_residual:
                    residual_job = ship_residuals()
                if idx == last_coded and not self.evaluate_unused_tail:
                    break
                lower = self.decoder_block[idx](res, fea_pred)
            else:
                lower = self.decoder_block[idx](fea_pred)
            del fea_pred
        lower = fea_pred = None
        for idx, lower_of_level, target_map in reversed(pending_occ):     # finest first
            code_occupancy(idx, lower_of_level, target_map)
        lower_of_level = None
        if n_flags > flags.size:
            raise RuntimeError('too many coded levels for the flag block')

        if tm is not None:
            tm['enc_enqueued'] = time.perf_counter()
        try:
            coord_bytes_list = pool.wait()                               # the only blocking point of the encoder
        except RuntimeError as e:
            # a coder refused its input (a zero probability, a symbol outside its own histogram): say what the streams held
            st['side'].synchronize()
            what = [f'occupancy level {i}: {m.size} symbols, prob16 min {int(q.min()) if q.size else -1} zeros {int((q == 0).sum())}'
                    for i, (m, q) in enumerate(occupancy_h)]
            if residual_job is not None:
                sym = residual_job[0]
                what.append(f'residuals: {sym.size} symbols in [{int(sym.min())}, {int(sym.max())}]')
            raise RuntimeError(f'{e}; job status words {getattr(pool, "last_failure", None)}; ' + '; '.join(what)) from e
        if pending_occ:                                                  # coded finest first: back to level order (coarse -> fine)
            coord_bytes_list.reverse()
            occupancy_h.reverse()
            pending_occ.clear()
        st['side'].synchronize()
        if 'occ' in st:
            st['occ'].synchronize()
        if tm is not None:
            tm['enc_synced'] = tm['enc_occupancy_coded'] = time.perf_counter()
        sym_h, job = residual_job
        if self.keep_symbols:      # test hook: what went into the coders
            self.last_symbols = {'residual': sym_h.copy(),
                                 'occupancy': np.concatenate([m for m, _ in occupancy_h]) if occupancy_h else np.zeros(0, np.uint8),
                                 'prob': np.concatenate([q for _, q in occupancy_h]) if occupancy_h else np.zeros(0, np.uint16),
                                 'sizes': [m.size for m, _ in occupancy_h]}

        with io.BytesIO() as bs:
            bs.write(int_to_bytes(bottom_map.level, 1))                                 # log2(bottom stride)
            bs.write(int_to_bytes(bottom_map.n, self.broadcast_shape_bytes))
            offset, cdf, payload = pool.histogram_result(job)
            self._write_coded(bs, sym_h.size, offset, cdf, payload, True)
            bs.write(int_to_bytes(len(coord_bytes_list), 1))
            BytesListUtils.concat_bytes_list(coord_bytes_list, bs)
            self.rans_encode_with_cdf(xyz_h, bs, 0)
            st['pinned'] = []
            if tm is not None:
                tm['enc_done'] = time.perf_counter()
            return bs.getvalue()

    # -- decompress -----------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def decompress(self, concat_bytes: bytes, cm: ME.CoordinateManager) -> ME.SparseTensor:
        dev = next(self.parameters()).device
        scale = float(self.bottleneck_scaler)
        tm = self.timing
        if tm is not None:
            tm['dec_t0'] = time.perf_counter()
            tm['dec_wait'] = tm['dec_host'] = 0.0
        with io.BytesIO(concat_bytes) as bs:
            bottom_level = bytes_to_int(bs.read(1))
            bottom_rows = bytes_to_int(bs.read(self.broadcast_shape_bytes))
            n_sym, sym_offset, sym_cdf, sym_payload = self._read_coded(bs, None)
            n_streams = bytes_to_int(bs.read(1))
            coord_bytes_list = BytesListUtils.split_bytes_list(None, n_streams, bs) if n_streams else []
            bottom_xyz, _ = self.rans_decode_with_cdf(bs, 0, 3)

        # the residual stream is decoded in the background, coarse levels first -- the order the loop below consumes it
        pool: CoderPool = self._overlap_state(dev)['pool']
        sym_t = torch.empty(n_sym * self.compressed_channels, dtype=torch.int32, pin_memory=True)
        progress = pool.table_decode(sym_payload, sym_t.numel(), sym_cdf, sym_offset, sym_t.numpy(),
                                     first_chunk=bottom_rows * self.compressed_channels)

        def residuals(first_row: int, rows: int) -> torch.Tensor:
            if first_row + rows > n_sym:
                raise ValueError('the bitstream holds fewer residual symbols than the pyramid needs')
            c = self.compressed_channels
            pool.need(progress, (first_row + rows) * c)
            t = sym_t[first_row * c: (first_row + rows) * c].to(dev, non_blocking=True).to(torch.float32).view(rows, c)
            return t if scale == 1.0 else t / scale

        coords = torch.zeros((bottom_rows, 4), dtype=torch.int32)
        coords[:, 1:] = torch.from_numpy(bottom_xyz) << bottom_level
        lower = ME.SparseTensor(residuals(0, bottom_rows), coordinates=coords.to(dev),
                                tensor_stride=1 << bottom_level, coordinate_manager=cm)
        if tm is not None:
            tm['dec_residual_decoded'] = time.perf_counter()
        used = bottom_rows

        for idx in range(len(self.residual_block) - 1, -1, -1):
            occ_net = self.hyper_decoder_coord[idx]
            cur_map = cm._map(lower.coordinate_map_key)
            if occ_net is not None:
                logits = occ_net(lower)
                ta = time.perf_counter()
                prob_d = ops.logit_to_prob16(logits.F.view(-1))
                if self.device_decoder:
                    raw = coord_bytes_list.pop(0)
                    mask, ones, status = ops.rans_binary_decode_dev(ops.stream_to_device(raw, dev), len(raw), prob_d)
                    count, ok = torch.cat((ones, status)).tolist()          # the level's one read-back: 8 bytes
                    if ok != 0:
                        raise ValueError('corrupt occupancy stream')
                    if tm is not None:
                        tm['dec_wait'] += time.perf_counter() - ta
                    gen_id = logits.coordinate_map_key.get_key()[1]
                    cur_map = cm._refine(cur_map, mask, gen_id + 'pruned', count_hint=int(count))
                    del logits
                    target_key = cur_map.key
                    fea_pred = self.hyper_decoder_fea[idx](lower, target_key)
                    if idx > self.skip_encoding_fea:
                        res = residuals(used, cur_map.n)
                        used += cur_map.n
                        lower = self.decoder_block[idx](res, fea_pred)
                    else:
                        lower = self.decoder_block[idx](fea_pred)
                    continue
                prob_t = torch.empty(prob_d.shape, dtype=prob_d.dtype, pin_memory=True)
                prob_t.copy_(prob_d, non_blocking=True)
                # the level's dependency: D2H of its probabilities.  An event, not a stream synchronise: with several frames in flight
                # on the stream (fastpcc_amd/serving.py) this frame waits for its copy, not for what other frames queued behind it
                copied = torch.cuda.Event()
                copied.record()
                copied.synchronize()
                tb = time.perf_counter()
                bits_t = torch.empty(prob_t.numel(), dtype=torch.uint8, pin_memory=True)
                stream = np.frombuffer(coord_bytes_list.pop(0), dtype=np.uint8)
                host_check(host().fpcc_rans_binary_decode(stream.ctypes.data, stream.size, prob_t.numpy().ctypes.data,
                                                          prob_t.numel(), bits_t.numpy().ctypes.data))
                mask = bits_t.to(dev, non_blocking=True)                 # H2D
                if tm is not None:
                    tm['dec_wait'] += tb - ta
                    tm['dec_host'] += time.perf_counter() - tb
                gen_id = logits.coordinate_map_key.get_key()[1]
                # the decoded mask is on the host: its popcount sizes the new map without a device read-back
                cur_map = cm._refine(cur_map, mask, gen_id + 'pruned', count_hint=int(np.count_nonzero(bits_t.numpy())))
                del logits
            target_key = cur_map.key
            fea_pred = self.hyper_decoder_fea[idx](lower, target_key)
            if idx > self.skip_encoding_fea:
                res = residuals(used, cur_map.n)
                used += cur_map.n
                lower = self.decoder_block[idx](res, fea_pred)
            else:
                lower = self.decoder_block[idx](fea_pred)
        if coord_bytes_list:
            raise ValueError('unused occupancy streams in the bitstream')
        pool.wait()
        if used != n_sym:
            raise ValueError('residual symbols left over in the bitstream')
        return lower
