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
        return self.compress_clouds(y_top, 1)[0]

    def _quiesce_pools(self) -> None:
        """after an exception inside compress / decompress: no job of the coder pools may outlive the frame (it reads and writes buffers
        the frame owns).  Copies still in flight are completed, encoder jobs that wait for a flag are released (their results are dropped),
        running jobs are waited for; errors of those jobs are swallowed -- the caller is already propagating the real one."""
        for dev, st in self._overlap.items():
            try:
                torch.cuda.synchronize(dev)
            except Exception:
                pass
            st['flags'].fill_(1)
            try:
                st['pool'].wait()
            except RuntimeError:
                pass
            st['pinned'] = []

    @torch.no_grad()
    def compress_clouds(self, y_top: ME.SparseTensor, n_clouds: int) -> List[bytes]:
        try:
            return self._compress_clouds(y_top, n_clouds)
        except BaseException:
            self._quiesce_pools()
            raise

    def _compress_clouds(self, y_top: ME.SparseTensor, n_clouds: int) -> List[bytes]:
        """`n_clouds` independent clouds on one coordinate manager (ME.CoordinateManager(clouds=n), rows cloud-major) through ONE
        traversal of the networks; every cloud gets the stream `compress` writes for it alone -- its own residual histogram, its own
        occupancy streams, its own header --, coded by its own jobs on the coder pool (the clouds' long passes run side by side).
        What the reference does with a list of clouds one cloud at a time (lossy_coord_v2/model.py:247-256)."""
        tm = self.timing
        if tm is not None:
            tm.clear()
            tm['enc_t0'] = time.perf_counter()
        cm = y_top.coordinate_manager
        B = int(n_clouds)
        if B != (cm._n_batch or 1) or (B > 1 and not cm.independent_clouds):
            raise ValueError('the coordinate manager was not made for this many independent clouds')
        *feas, bottom = self.encoder(y_top, 1)
        del y_top
        scale = float(self.bottleneck_scaler)
        bottom_f = bottom.F
        st = self._overlap_state(bottom_f.device)
        st['pinned'] = []
        st['flags'].zero_()
        pool: CoderPool = st['pool']
        flags = st['flags'].numpy().view(np.uint32)
        n_flags = 0
        edges_of = (lambda m: cm.batch_offsets(m)) if B > 1 else (lambda m: [0, m.n])
        bottom_map = cm._map(bottom.coordinate_map_key)
        residual_syms = [ops.quantize_symbols_(bottom_f, scale)]         # rounds bottom_f in place
        residual_edges = [edges_of(bottom_map)]                          # per residual level: its clouds' row ranges
        occupancy_h: List[Tuple[np.ndarray, np.ndarray, List[int]]] = []  # (mask u8, prob u16, cloud ranges) per coded level, host side
        lower = bottom
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
            """all residual symbols in ONE copy, cloud-major (a cloud's stream holds its rows of every level, coarse level first),
            one flag, one histogram job per cloud"""
            nonlocal n_flags
            if B == 1:
                pieces, sizes = residual_syms, [sum(t.numel() for t in residual_syms)]
            else:
                pieces = [t[e[c]:e[c + 1]] for c in range(B) for t, e in zip(residual_syms, residual_edges)]
                sizes = [sum(e[c + 1] - e[c] for e in residual_edges) * self.compressed_channels for c in range(B)]
            (sym_h,) = self._ship(st, [torch.cat(pieces)], n_flags)
            jobs, at = [], 0
            for size in sizes:
                part = sym_h[at: at + size]
                jobs.append((part, pool.histogram_encode(part, None, flags[n_flags:n_flags + 1])))
                at += size
            n_flags += 1
            return jobs

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
            cand = [8 * e for e in edges_of(target_map.parent)]          # the candidates of a cloud: 8 per row of the level above
            for c in range(B):
                pool.binary_encode(mask_h[cand[c]:cand[c + 1]], prob_h[cand[c]:cand[c + 1]], flags[n_flags:n_flags + 1])
            n_flags += 1
            occupancy_h.append((mask_h, prob_h, cand))

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
                residual_edges.append(edges_of(target_map))
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
            streams = pool.wait()                                        # the only blocking point of the encoder
        except RuntimeError as e:
            # A coder refused its input (a zero probability, a symbol outside its own histogram).  The inputs of a job are what a
            # stream-ordered copy left in pinned memory before the job's flag; with every copy complete (synchronise) they are
            # coded again right here, inline.  If that succeeds the refusal was a hand-over fault -- counted in `handover_retries`
            # (never observed since the counter exists; bench.py reports it) --, otherwise the symbols themselves are not codable.
            st['side'].synchronize()
            if 'occ' in st:
                st['occ'].synchronize()
            what = [f'occupancy level {i}: {m.size} symbols, prob16 min {int(q.min()) if q.size else -1} zeros {int((q == 0).sum())}'
                    for i, (m, q, _) in enumerate(occupancy_h)]
            for part, _ in residual_job or []:
                what.append(f'residuals: {part.size} symbols in [{int(part.min())}, {int(part.max())}]')
            note = f'{e}; job status words {getattr(pool, "last_failure", None)}; ' + '; '.join(what)
            try:
                streams = []
                for m, q, cand in occupancy_h:
                    for c in range(B):
                        streams.append(self.binary_rans_coder.encode(m[None, cand[c]:cand[c + 1]].astype(bool),
                                                                     q[None, cand[c]:cand[c + 1]].astype(np.uint32))[0])
                redo = CoderPool(1)
                residual_job = [(part, redo.histogram_encode(part, None)) for part, _ in residual_job]
                redo.wait()
                redo.close()
            except RuntimeError:
                raise RuntimeError(note) from e
            GeoLosslessEntropyModel.handover_retries += 1
            import warnings
            warnings.warn('coder pool job failed and succeeded when repeated after a synchronise: ' + note)
        n_levels = len(occupancy_h)
        by_level = [streams[l * B:(l + 1) * B] for l in range(n_levels)]
        if pending_occ:                                                  # coded finest first: back to level order (coarse -> fine)
            by_level.reverse()
            occupancy_h.reverse()
            pending_occ.clear()
        st['side'].synchronize()
        if 'occ' in st:
            st['occ'].synchronize()
        if tm is not None:
            tm['enc_synced'] = tm['enc_occupancy_coded'] = time.perf_counter()
        if self.keep_symbols:      # test hook: what went into the coders
            kept = []
            for c in range(B):
                masks = [m[cand[c]:cand[c + 1]] for m, _, cand in occupancy_h]
                probs = [q[cand[c]:cand[c + 1]] for _, q, cand in occupancy_h]
                kept.append({'residual': residual_job[c][0].copy(),
                             'occupancy': np.concatenate(masks) if masks else np.zeros(0, np.uint8),
                             'prob': np.concatenate(probs) if probs else np.zeros(0, np.uint16),
                             'sizes': [m.size for m in masks]})
            self.last_symbols = kept[0] if B == 1 else kept

        out = []
        bottom_edges = residual_edges[0]
        for c in range(B):
            with io.BytesIO() as bs:
                bs.write(int_to_bytes(bottom_map.level, 1))                             # log2(bottom stride)
                bs.write(int_to_bytes(bottom_edges[c + 1] - bottom_edges[c], self.broadcast_shape_bytes))
                sym_h, job = residual_job[c]
                offset, cdf, payload = pool.histogram_result(job)
                self._write_coded(bs, sym_h.size, offset, cdf, payload, True)
                bs.write(int_to_bytes(n_levels, 1))
                BytesListUtils.concat_bytes_list([level[c] for level in by_level], bs)
                self.rans_encode_with_cdf(xyz_h[bottom_edges[c]:bottom_edges[c + 1]], bs, 0)
                out.append(bs.getvalue())
        st['pinned'] = []
        if tm is not None:
            tm['enc_done'] = time.perf_counter()
        return out

    handover_retries = 0      # process-wide: coder-pool jobs that failed and succeeded when repeated after a synchronise

    # -- decompress -----------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def decompress(self, concat_bytes: bytes, cm: ME.CoordinateManager) -> ME.SparseTensor:
        return self.decompress_clouds([concat_bytes], cm)

    @torch.no_grad()
    def decompress_clouds(self, streams: List[bytes], cm: ME.CoordinateManager) -> ME.SparseTensor:
        try:
            return self._decompress_clouds(streams, cm)
        except BaseException:
            self._quiesce_pools()          # the background residual decode writes into a buffer this frame owns
            raise

    def _decompress_clouds(self, streams: List[bytes], cm: ME.CoordinateManager) -> ME.SparseTensor:
        """The streams of `len(streams)` independent clouds (each what `compress` writes) decoded in ONE traversal of the networks on
        `cm` (made with clouds=len(streams)): per occupancy level one device->host copy of the probabilities of all clouds, the
        clouds' streams decoded side by side on the coder pool, one host->device copy of the mask.  Returns the reconstructed top
        features of the union (rows cloud-major; cm.batch_offsets gives the ranges)."""
        dev = next(self.parameters()).device
        scale = float(self.bottleneck_scaler)
        B = len(streams)
        if B != (cm._n_batch or 1) or (B > 1 and not cm.independent_clouds):
            raise ValueError('the coordinate manager was not made for this many independent clouds')
        tm = self.timing
        if tm is not None:
            tm['dec_t0'] = time.perf_counter()
            tm['dec_wait'] = tm['dec_host'] = 0.0
        parsed = []
        for data in streams:
            with io.BytesIO(data) as bs:
                bottom_level = bytes_to_int(bs.read(1))
                bottom_rows = bytes_to_int(bs.read(self.broadcast_shape_bytes))
                n_sym, sym_offset, sym_cdf, sym_payload = self._read_coded(bs, None)
                n_streams = bytes_to_int(bs.read(1))
                coord_bytes_list = BytesListUtils.split_bytes_list(None, n_streams, bs) if n_streams else []
                bottom_xyz, _ = self.rans_decode_with_cdf(bs, 0, 3)
            parsed.append((bottom_level, bottom_rows, n_sym, sym_offset, sym_cdf, sym_payload, coord_bytes_list, bottom_xyz))
        bottom_level = parsed[0][0]
        if any(p[0] != bottom_level for p in parsed):
            raise ValueError('the clouds of one traversal must share the depth of the pyramid')

        # the residual streams are decoded in the background, coarse levels first -- the order the loop below consumes them
        pool: CoderPool = self._overlap_state(dev)['pool']
        c_ch = self.compressed_channels
        sym_t, progress = [], []
        for _, rows, n_sym, sym_offset, sym_cdf, sym_payload, _, _ in parsed:
            t = torch.empty(n_sym * c_ch, dtype=torch.int32, pin_memory=True)
            progress.append(pool.table_decode(sym_payload, t.numel(), sym_cdf, sym_offset, t.numpy(), first_chunk=rows * c_ch))
            sym_t.append(t)
        used = [0] * B

        def residuals(rows_of_cloud: List[int]) -> torch.Tensor:
            """the next rows_of_cloud[c] residual rows of every cloud, cloud-major, on the device"""
            for c, rows in enumerate(rows_of_cloud):
                if used[c] + rows > parsed[c][2]:
                    raise ValueError('the bitstream holds fewer residual symbols than the pyramid needs')
            if B == 1:
                rows = rows_of_cloud[0]
                pool.need(progress[0], (used[0] + rows) * c_ch)
                t = sym_t[0][used[0] * c_ch: (used[0] + rows) * c_ch].to(dev, non_blocking=True)
            else:
                t = torch.empty(sum(rows_of_cloud) * c_ch, dtype=torch.int32, device=dev)
                at = 0
                for c, rows in enumerate(rows_of_cloud):
                    pool.need(progress[c], (used[c] + rows) * c_ch)
                    t[at: at + rows * c_ch].copy_(sym_t[c][used[c] * c_ch: (used[c] + rows) * c_ch], non_blocking=True)
                    at += rows * c_ch
            for c, rows in enumerate(rows_of_cloud):
                used[c] += rows
            t = t.to(torch.float32).view(-1, c_ch)
            return t if scale == 1.0 else t / scale

        bottom_rows = [p[1] for p in parsed]
        coords = torch.zeros((sum(bottom_rows), 4), dtype=torch.int32)
        at = 0
        for c, p in enumerate(parsed):
            coords[at: at + p[1], 0] = c
            coords[at: at + p[1], 1:] = torch.from_numpy(p[7]) << bottom_level
            at += p[1]
        lower = ME.SparseTensor(residuals(bottom_rows), coordinates=coords.to(dev),
                                tensor_stride=1 << bottom_level, coordinate_manager=cm)
        if B > 1:
            # the rows of the bottom map are the clouds' bottom rows in stream order (each cloud's are Morton-sorted and distinct, the
            # cloud index is the key's top): its ranges are known without a read-back
            bm = cm._map(lower.coordinate_map_key)
            if bm.n != sum(bottom_rows):
                raise ValueError('duplicate bottom coordinates in a stream')
            bm.edges = ME._edges_of(bottom_rows)
        if tm is not None:
            tm['dec_residual_decoded'] = time.perf_counter()
        occupancy = [p[6] for p in parsed]

        for idx in range(len(self.residual_block) - 1, -1, -1):
            occ_net = self.hyper_decoder_coord[idx]
            cur_map = cm._map(lower.coordinate_map_key)
            if occ_net is not None:
                logits = occ_net(lower)
                ta = time.perf_counter()
                prob_d = ops.logit_to_prob16(logits.F.view(-1))
                if self.device_decoder:
                    if B != 1:
                        raise NotImplementedError('the device-side occupancy decoder takes one cloud')
                    raw = occupancy[0].pop(0)
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
                        res = residuals([cur_map.n])
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
                prob_h, bits_h = prob_t.numpy().view(np.uint16), bits_t.numpy()
                cand = [8 * e for e in (cm.batch_offsets(cur_map) if B > 1 else [0, cur_map.n])]
                raws = [np.frombuffer(occupancy[c].pop(0), dtype=np.uint8) for c in range(B)]
                # clouds 1 .. B-1 on the pool's threads, cloud 0 on this one
                waits = [pool.binary_decode(raws[c], prob_h[cand[c]:cand[c + 1]], bits_h[cand[c]:cand[c + 1]]) for c in range(1, B)]
                host_check(host().fpcc_rans_binary_decode(raws[0].ctypes.data, raws[0].size, prob_h[cand[0]:].ctypes.data,
                                                          cand[1] - cand[0], bits_h.ctypes.data))
                for w in waits:
                    pool.need(w, 1)
                mask = bits_t.to(dev, non_blocking=True)                 # H2D
                if tm is not None:
                    tm['dec_wait'] += tb - ta
                    tm['dec_host'] += time.perf_counter() - tb
                gen_id = logits.coordinate_map_key.get_key()[1]
                # the decoded mask is on the host: its popcount sizes the new map (and its clouds' ranges) without a device read-back
                counts = [int(np.count_nonzero(bits_h[cand[c]:cand[c + 1]])) for c in range(B)]
                cur_map = cm._refine(cur_map, mask, gen_id + 'pruned', count_hint=sum(counts), cloud_counts=counts if B > 1 else None)
                del logits
            target_key = cur_map.key
            fea_pred = self.hyper_decoder_fea[idx](lower, target_key)
            if idx > self.skip_encoding_fea:
                e = cm.batch_offsets(cur_map) if B > 1 else [0, cur_map.n]
                res = residuals([e[c + 1] - e[c] for c in range(B)])
                lower = self.decoder_block[idx](res, fea_pred)
            else:
                lower = self.decoder_block[idx](fea_pred)
        if any(occupancy):
            raise ValueError('unused occupancy streams in the bitstream')
        pool.wait()
        if any(used[c] != parsed[c][2] for c in range(B)):
            raise ValueError('residual symbols left over in the bitstream')
        return lower
