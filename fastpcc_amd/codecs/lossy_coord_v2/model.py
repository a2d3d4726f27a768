"""lossy_coord_v2 point-cloud geometry codec: encode/decode entry points with the module tree (state_dict keys), the
method names and the frame layout of `PCC` in /root/reference/models/convolutional/lossy_coord_v2/model.py:23-288.

    compress(batched_coord)   -> bytes      u16 x0,y0,z0 | u24 target point count per pruning stage | lossless payload
    decompress(bytes)         -> int32 [N, 3]
    test_forward(PCData)      -> dict with the reconstructed cloud, the bytes and the two wall-clock times, timed the
                                 way the reference's Timer blocks are (device synchronised at both ends).

Training (`train_forward`) is not part of this inference build.
"""
import io
import time
from typing import List, Optional, Union

import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import engine as ME
from ... import hipops
from ...data import PCData
from ...evaluators import PCCEvaluator
from ..geo_lossl_em import GeoLosslessEntropyModel
from .layers import Decoder, DecoderGeoLossl, Encoder, EncoderGeoLossl, HyperDecoderGenUpsample, \
    HyperDecoderUpsample, ResidualGeoLossl
from .model_config import ModelConfig


class PCC(nn.Module):

    @staticmethod
    def params_divider(s: str) -> int:
        return 1 if 'bottom_fea_entropy_model' in s else 0

    def __init__(self, cfg: ModelConfig):
        super().__init__()
        self.cfg = cfg
        ME.set_sparse_tensor_operation_mode(ME.SparseTensorOperationMode.SHARE_COORDINATE_MANAGER)
        self.evaluator = PCCEvaluator()
        if len(cfg.compressed_channels) != len(cfg.geo_lossl_channels) or \
                len(cfg.geo_lossl_if_sample) != len(cfg.geo_lossl_channels) - 1 or \
                cfg.compressed_channels[-1] != cfg.geo_lossl_channels[-1]:
            raise ValueError('inconsistent geo_lossl_* / compressed_channels configuration')
        ch = cfg.geo_lossl_channels
        region, act = cfg.conv_region_type, cfg.activation

        self.encoder = Encoder(1, cfg.encoder_channels, cfg.adaptive_pruning, cfg.adaptive_pruning_scaler, region, act)
        self.decoder = Decoder(ch[0], cfg.decoder_channels, region, act)
        self.em_lossless_based = GeoLosslessEntropyModel(
            cfg.compressed_channels[0], cfg.bottleneck_process, cfg.bottleneck_scaler, cfg.skip_encoding_fea,
            encoder=EncoderGeoLossl(ch[:-1], ch, cfg.geo_lossl_if_sample, region, act, cfg.bottleneck_value_bound,
                                    cfg.skip_encoding_fea),
            residual_block=ResidualGeoLossl(ch[:-1], cfg.compressed_channels[:-1], region, act,
                                            cfg.bottleneck_value_bound, cfg.skip_encoding_fea),
            decoder_block=DecoderGeoLossl(cfg.compressed_channels[:-1], ch[:-1], ch[:-1], region, act,
                                          cfg.skip_encoding_fea),
            hyper_decoder_coord=HyperDecoderGenUpsample(ch[1:], cfg.geo_lossl_if_sample, region, act),
            hyper_decoder_fea=HyperDecoderUpsample(ch[1:], ch[:-1], cfg.geo_lossl_if_sample, region, act))

    # ---------------------------------------------------------------------------------------------------------------
    def forward(self, pc_data: PCData):
        if self.training:
            return self.train_forward(pc_data.xyz, pc_data.training_step, pc_data.batch_size)
        if pc_data.batch_size != 1:
            raise ValueError('Only supports batch size == 1 during testing.')
        return self.test_forward(pc_data)

    def set_global_cm(self, clouds: Optional[int] = None) -> ME.CoordinateManager:
        ME.clear_global_coordinate_manager()
        cm = ME.CoordinateManager(D=3, clouds=clouds)
        ME.set_global_coordinate_manager(cm)
        return cm

    def get_sparse_pc(self, xyz: torch.Tensor, clouds: Optional[int] = None) -> ME.SparseTensor:
        """All-ones 1-channel tensor on the input voxels.  The engine keeps every map in Morton order, which is what the
        reference obtains by sorting explicitly before building the tensor (model.py:140-142).  clouds: the batch column of `xyz`
        numbers that many independent clouds coded in one traversal (compress_many)."""
        cm = self.set_global_cm(clouds)
        ones = torch.ones((xyz.shape[0], 1), dtype=torch.float32, device=xyz.device)
        pc = ME.SparseTensor(features=ones, coordinates=xyz, tensor_stride=[1] * 3, coordinate_manager=cm,
                             quantization_mode=ME.SparseTensorQuantizationMode.UNWEIGHTED_AVERAGE)
        pc._fpcc_all_ones = True        # lets the first 3x3x3 layer work from neighbour-presence masks (engine.py)
        # all coarser maps the encoder + lossless pyramid will ask for, built while the stream is still empty
        cm.build_pyramid(pc.coordinate_map_key, len(self.cfg.encoder_channels) - 1 + sum(self.cfg.geo_lossl_if_sample))
        return pc

    def train_forward(self, batched_coord: torch.Tensor, training_step: int, batch_size: int) -> dict:
        """rate + distortion objective of one batch (model.py:156-191): returns {'loss': tensor with the autograd graph,
        every other term detached}.  The reference converts the other terms to floats right here (`.item()`, one device
        synchronisation per term BEFORE the backward pass is queued); they stay 0-dim device tensors and whoever logs them
        (fastpcc_amd.train.Trainer.step) reads them after the update has been enqueued."""
        sparse_pc = self.get_sparse_pc(batched_coord)
        feature, points_num_list = self.encoder(sparse_pc)
        bottleneck_feature, loss_dict = self.em_lossless_based(feature, batch_size)
        for k, v in self.decoder(bottleneck_feature, points_num_list, sparse_pc.coordinate_map_key).items():
            loss_dict[k] = loss_dict[k] + v if k in loss_dict else v
        cfg = self.cfg
        if training_step < cfg.warmup_fea_loss_steps:
            step = (cfg.warmup_fea_loss_factor - cfg.bits_loss_factor) / cfg.warmup_fea_loss_steps
            fea_factor = cfg.warmup_fea_loss_factor - step * training_step if cfg.linear_warmup else cfg.warmup_fea_loss_factor
        else:
            fea_factor = cfg.bits_loss_factor
        for key in loss_dict:
            if key.endswith('bits_loss'):
                loss_dict[key] = loss_dict[key] * (fea_factor if 'fea' in key else cfg.bits_loss_factor)
            if key.startswith('coord_recon_loss'):          # (sic) the reference's prefix test, model.py:187
                loss_dict[key] = loss_dict[key] * cfg.coord_recon_loss_factor
        loss_dict['loss'] = sum(loss_dict.values())
        for key in loss_dict:
            if key != 'loss':
                loss_dict[key] = loss_dict[key].detach()
        return loss_dict

    # ---------------------------------------------------------------------------------------------------------------
    @hipops.no_gc_pause
    @torch.no_grad()
    def compress(self, batched_coord: torch.Tensor) -> bytes:
        if not batched_coord.is_cuda:
            raise RuntimeError('compress() runs on the GPU; move the coordinates there first')
        coord_offset = batched_coord.amin(0)[1:]        # reduce the contiguous [n, 4] tensor, then drop the batch column
        sparse_pc = self.get_sparse_pc((batched_coord - F.pad(coord_offset, (1, 0))).contiguous())
        feature, points_num_list = self.encoder(sparse_pc)
        em_bytes = self.em_lossless_based.compress(feature, 1)
        return self._header(coord_offset.tolist(), None if points_num_list is None else [counts[0] for counts in points_num_list]) + em_bytes

    def _header(self, coord_offset: List[int], counts: Optional[List[int]]) -> bytes:
        head = bytes([hipops.numerics_version()]) if self.cfg.numerics_version_in_header else b''
        head += b''.join(int(v).to_bytes(2, 'little', signed=False) for v in coord_offset)
        if self.cfg.adaptive_pruning:
            head += b''.join(int(n).to_bytes(3, 'little', signed=False) for n in counts)
        return head

    @hipops.no_gc_pause
    @torch.no_grad()
    def compress_many(self, clouds: List[torch.Tensor]) -> List[bytes]:
        """Independent clouds (each int32 [n, 4], batch column 0) -> the stream `compress` writes for each of them, from ONE traversal
        of the networks over their union: the clouds become the samples of one batch (rows of different clouds never neighbour), every
        layer is launched once over all of them, and each cloud's symbols go to its own coder jobs.  The launches of a cloud's small
        pyramid levels -- latency-bound alone -- carry every cloud's rows.  What the reference does with the partitions of a large
        cloud one at a time (model.py:247-256); bytes are identical to coding each cloud alone (tests/test_gpu_codec_many.py)."""
        if len(clouds) == 1:
            return [self.compress(clouds[0])]
        if not clouds or not all(c.is_cuda for c in clouds):
            raise RuntimeError('compress_many() takes a non-empty list of GPU tensors')
        B = len(clouds)
        offsets = torch.stack([c.amin(0) for c in clouds])              # [B, 4]; the batch column of every cloud is 0
        shift = offsets.clone()
        shift[:, 0] = -torch.arange(B, device=shift.device, dtype=shift.dtype)          # cloud b becomes sample b of the batch
        sparse_pc = self.get_sparse_pc(torch.cat([c - shift[b] for b, c in enumerate(clouds)]), clouds=B)
        feature, points_num_list = self.encoder(sparse_pc)
        em_bytes = self.em_lossless_based.compress_clouds(feature, B)
        offsets = offsets[:, 1:].tolist()
        return [self._header(offsets[b], None if points_num_list is None else [counts[b] for counts in points_num_list]) + em_bytes[b]
                for b in range(B)]

    @hipops.no_gc_pause
    @torch.no_grad()
    def decompress_many(self, streams: List[bytes]) -> List[torch.Tensor]:
        """inverse of compress_many: the clouds' streams decoded in one traversal; -> one int32 [N_b, 3] tensor per cloud"""
        if len(streams) == 1:
            return [self.decompress(streams[0])]
        dev = next(self.parameters()).device
        B = len(streams)
        offsets, targets, payloads = [], [], []
        for data in streams:
            offset, counts, em_bytes = self._parse_header(data)
            offsets.append(offset)
            targets.append(counts)
            payloads.append(em_bytes)
        offset_t = torch.tensor(offsets, dtype=torch.int32, device=dev)    # made before anything is queued (pageable memory)
        points_num_list = None
        if self.cfg.adaptive_pruning:
            points_num_list = [[targets[b][s][0] for b in range(B)] for s in range(len(targets[0]))]
        fea_recon = self.em_lossless_based.decompress_clouds(payloads, self.set_global_cm(B))
        return self.decoder(fea_recon, points_num_list, offset_t)

    # partitions of one list are coded together up to this many voxels per traversal (activations of every level stay resident)
    MANY_MAX_VOXELS = 6_000_000

    def _groups(self, sizes: List[int]) -> List[List[int]]:
        """consecutive list entries per traversal: as many as fit MANY_MAX_VOXELS (and the engine's 64 clouds)"""
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

    def compress_partitions(self, batched_coord: List[torch.Tensor]) -> bytes:
        # element 0 is the unpartitioned cloud (model.py:249-250)
        parts = list(batched_coord[1:])
        coded: List[bytes] = []
        for g in self._groups([p.shape[0] for p in parts]):
            coded.extend(self.compress_many([parts[i] for i in g]))
        return b''.join(len(s).to_bytes(3, 'little', signed=False) + s for s in coded)

    def _parse_header(self, compressed_bytes: bytes):
        with io.BytesIO(compressed_bytes) as bs:
            if self.cfg.numerics_version_in_header:
                written_by = bs.read(1)[0]
                if written_by != hipops.numerics_version():
                    raise ValueError(f'stream written with numerics version {written_by}, this build decodes version '
                                     f'{hipops.numerics_version()} (the fp32 summation orders are part of the format)')
            coord_offset = [int.from_bytes(bs.read(2), 'little', signed=False) for _ in range(3)]
            points_num_list = None
            if self.cfg.adaptive_pruning:
                points_num_list = [[int.from_bytes(bs.read(3), 'little', signed=False)]
                                   for _ in range(len(self.encoder.blocks) - 1)]
            return coord_offset, points_num_list, bs.read()

    @hipops.no_gc_pause
    @torch.no_grad()
    def decompress(self, compressed_bytes: bytes) -> torch.Tensor:
        dev = next(self.parameters()).device
        coord_offset, points_num_list, em_bytes = self._parse_header(compressed_bytes)
        # made before anything is queued: a host-to-device copy from pageable memory waits for the stream
        offset = torch.tensor(coord_offset, dtype=torch.int32, device=dev)
        fea_recon = self.em_lossless_based.decompress(em_bytes, self.set_global_cm())
        return self.decoder(fea_recon, points_num_list, offset)

    def decompress_partitions(self, concat_bytes: bytes) -> torch.Tensor:
        streams = []
        with io.BytesIO(concat_bytes) as bs:
            while bs.tell() != len(concat_bytes):
                length = int.from_bytes(bs.read(3), 'little', signed=False)
                streams.append(bs.read(length))
        out: List[torch.Tensor] = []
        # a partition's size before it is decoded: the pruning target of the finest stage in its header (adaptive pruning), else the
        # stream length at one bit per voxel
        if self.cfg.adaptive_pruning:
            sizes = [self._parse_header(s)[1][0][0] for s in streams]
        else:
            sizes = [len(s) * 8 for s in streams]
        for g in self._groups(sizes):
            out.extend(self.decompress_many([streams[i] for i in g]))
        return torch.cat(out, 0)

    # ---------------------------------------------------------------------------------------------------------------
    def test_forward(self, pc_data: PCData) -> dict:
        whole = isinstance(pc_data.xyz, torch.Tensor)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        compressed_bytes = self.compress(pc_data.xyz) if whole else self.compress_partitions(pc_data.xyz)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        ME.clear_global_coordinate_manager()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        coord_recon = self.decompress(compressed_bytes) if whole else self.decompress_partitions(compressed_bytes)
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        ME.clear_global_coordinate_manager()
        if pc_data.inv_transform is not None:
            inv = pc_data.inv_transform[0].to(coord_recon.device)
            pred_xyz = coord_recon * inv[3] + inv[None, :3]
            compressed_bytes = pc_data.inv_transform[0].numpy().astype('<f4').tobytes() + compressed_bytes
        else:
            pred_xyz = coord_recon
        org = pc_data.xyz if whole else pc_data.xyz[0]
        n_org = pc_data.org_points_num[0] if pc_data.org_points_num else org.shape[0]
        ret = {'pred': pred_xyz, 'compressed_bytes': compressed_bytes, 'bpp': 8 * len(compressed_bytes) / n_org,
               'encode time': t1 - t0, 'decode time': t3 - t2}
        if pc_data.resolution is not None:
            # the reference hands the reconstruction to PCCEvaluator (model.py:215-228), which runs pc_error on files;
            # here the D1 distortion is computed on the device from the tensors at hand
            self.evaluator.log(pred=coord_recon, org_points_num=n_org, compressed_bytes=compressed_bytes,
                               file_path=pc_data.file_path[0] if pc_data.file_path else f'sample{len(self.evaluator.file_path_to_info)}',
                               resolution=pc_data.resolution[0], results_dir=pc_data.results_dir,
                               extra_info_dict={'encode time': t1 - t0, 'decode time': t3 - t2},
                               org_xyz=org[:, 1:] if pc_data.inv_transform is None else None)
        return ret
