"""Joint geometry + colour codec: module tree and frame layout of `PCC` in
/root/reference/models/convolutional/lossy_coord_lossy_color/model.py:23-314 (inference).  Differences to lossy_coord_v2:
4-channel input (R, G, B in [0,1] and a constant 2), a 3-stage encoder to stride 4, a two-stage generative decoder whose
last stage also predicts the colours, and every level of the lossless coder carries residual features."""
import io
import time
from typing import List, Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import engine as ME
from ... import hipops
from ...data import PCData
from ..geo_lossl_em import GeoLosslessEntropyModel
from .layers import Decoder, DecoderGeoLossl, Encoder, EncoderGeoLossl, HyperDecoderGenUpsample, HyperDecoderUpsample, \
    ResidualGeoLossl
from .model_config import ModelConfig


class PCC(nn.Module):

    @staticmethod
    def params_divider(s: str) -> int:
        return 1 if 'bottom_fea_entropy_model' in s else 0

    def __init__(self, cfg: ModelConfig):
        super().__init__()
        self.cfg = cfg
        ME.set_sparse_tensor_operation_mode(ME.SparseTensorOperationMode.SHARE_COORDINATE_MANAGER)
        ch = cfg.geo_lossl_channels
        region, act = cfg.conv_region_type, cfg.activation
        self.encoder = Encoder(4, ch[0], cfg.encoder_channels, cfg.adaptive_pruning, cfg.adaptive_pruning_scaler_train,
                               cfg.adaptive_pruning_scaler_test, region, act)
        self.decoder = Decoder(ch[0], 3, cfg.decoder_channels, region, act, cfg.use_yuv_loss)
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

    def forward(self, pc_data: PCData):
        if self.training:
            raise NotImplementedError('training is not part of this inference build')
        if pc_data.batch_size != 1:
            raise ValueError('Only supports batch size == 1 during testing.')
        return self.test_forward(pc_data)

    def set_global_cm(self, clouds: Optional[int] = None) -> ME.CoordinateManager:
        ME.clear_global_coordinate_manager()
        cm = ME.CoordinateManager(D=3, clouds=clouds)
        ME.set_global_coordinate_manager(cm)
        return cm

    def get_sparse_pc(self, xyz: torch.Tensor, color: torch.Tensor, clouds: Optional[int] = None) -> ME.SparseTensor:
        cm = self.set_global_cm(clouds)
        feats = torch.cat((color.to(torch.float32) / 255, torch.full((color.shape[0], 1), 2.0, device=color.device)), 1)
        pc = ME.SparseTensor(features=feats, coordinates=xyz, tensor_stride=[1] * 3, coordinate_manager=cm,
                             quantization_mode=ME.SparseTensorQuantizationMode.UNWEIGHTED_AVERAGE)
        cm.build_pyramid(pc.coordinate_map_key, len(self.cfg.encoder_channels) - 1 + sum(self.cfg.geo_lossl_if_sample))
        return pc

    @hipops.no_gc_pause
    @torch.no_grad()
    def compress(self, batched_coord: torch.Tensor, batched_color: torch.Tensor) -> bytes:
        if not batched_coord.is_cuda:
            raise RuntimeError('compress() runs on the GPU; move the inputs there first')
        coord_offset = batched_coord.amin(0)[1:]
        sparse_pc = self.get_sparse_pc((batched_coord - F.pad(coord_offset, (1, 0))).contiguous(), batched_color)
        feature, points_num_list = self.encoder(sparse_pc)
        em_bytes = self.em_lossless_based.compress(feature, 1)
        return self._header(coord_offset.tolist(), None if points_num_list is None else [counts[0] for counts in points_num_list]) + em_bytes

    def _header(self, coord_offset: List[int], counts: Optional[List[int]]) -> bytes:
        head = b''.join(int(v).to_bytes(2, 'little', signed=False) for v in coord_offset)
        if self.cfg.adaptive_pruning:
            head += b''.join(int(n).to_bytes(3, 'little', signed=False) for n in counts)
        return head

    @hipops.no_gc_pause
    @torch.no_grad()
    def compress_many(self, clouds: List[torch.Tensor], colors: List[torch.Tensor]) -> List[bytes]:
        """Independent coloured clouds -> the stream `compress` writes for each, from ONE traversal of the networks over their union
        (see lossy_coord_v2.PCC.compress_many; the reference codes the partitions of a large cloud one at a time, model.py:277-288)."""
        if len(clouds) != len(colors) or not clouds:
            raise ValueError('one colour tensor per cloud')
        if len(clouds) == 1:
            return [self.compress(clouds[0], colors[0])]
        B = len(clouds)
        offsets = torch.stack([c.amin(0) for c in clouds])
        shift = offsets.clone()
        shift[:, 0] = -torch.arange(B, device=shift.device, dtype=shift.dtype)
        sparse_pc = self.get_sparse_pc(torch.cat([c - shift[b] for b, c in enumerate(clouds)]), torch.cat(list(colors)), clouds=B)
        feature, points_num_list = self.encoder(sparse_pc)
        em_bytes = self.em_lossless_based.compress_clouds(feature, B)
        offsets = offsets[:, 1:].tolist()
        return [self._header(offsets[b], None if points_num_list is None else [counts[b] for counts in points_num_list]) + em_bytes[b]
                for b in range(B)]

    @hipops.no_gc_pause
    @torch.no_grad()
    def decompress_many(self, streams: List[bytes]) -> List[Tuple[torch.Tensor, torch.Tensor]]:
        if len(streams) == 1:
            return [self.decompress(streams[0])]
        dev = next(self.parameters()).device
        B = len(streams)
        parsed = [self._parse_header(s) for s in streams]
        offset = torch.tensor([p[0] for p in parsed], dtype=torch.int32, device=dev)      # before anything is queued
        points_num_list = None
        if self.cfg.adaptive_pruning:
            points_num_list = [[parsed[b][1][s][0] for b in range(B)] for s in range(len(parsed[0][1]))]
        cm = self.set_global_cm(B)
        fea_recon = self.em_lossless_based.decompress_clouds([p[2] for p in parsed], cm)
        out = self.decoder(fea_recon, points_num_list)
        edges = cm.batch_offsets(cm._map(out.coordinate_map_key))
        xyz, rgb = out.C[:, 1:], out.F.round_()
        return [(xyz[a:b] + offset[c], rgb[a:b]) for c, (a, b) in enumerate(zip(edges[:-1], edges[1:]))]

    MANY_MAX_VOXELS = 6_000_000       # partitions of one list coded per traversal (see lossy_coord_v2.PCC)

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

    def compress_partitions(self, batched_coord: List[torch.Tensor], batched_color: List[torch.Tensor]) -> bytes:
        coords, colors = list(batched_coord[1:]), list(batched_color[1:])      # element 0 is the unpartitioned cloud
        coded: List[bytes] = []
        for g in self._groups([c.shape[0] for c in coords]):
            coded.extend(self.compress_many([coords[i] for i in g], [colors[i] for i in g]))
        return b''.join(len(s).to_bytes(3, 'little', signed=False) + s for s in coded)

    def _parse_header(self, compressed_bytes: bytes):
        with io.BytesIO(compressed_bytes) as bs:
            coord_offset = [int.from_bytes(bs.read(2), 'little', signed=False) for _ in range(3)]
            points_num_list = None
            if self.cfg.adaptive_pruning:
                points_num_list = [[int.from_bytes(bs.read(3), 'little', signed=False)]
                                   for _ in range(len(self.cfg.decoder_channels))]
            return coord_offset, points_num_list, bs.read()

    @hipops.no_gc_pause
    @torch.no_grad()
    def decompress(self, compressed_bytes: bytes) -> Tuple[torch.Tensor, torch.Tensor]:
        dev = next(self.parameters()).device
        coord_offset, points_num_list, em_bytes = self._parse_header(compressed_bytes)
        offset = torch.tensor(coord_offset, dtype=torch.int32, device=dev)      # before anything is queued (pageable H2D waits)
        fea_recon = self.em_lossless_based.decompress(em_bytes, self.set_global_cm())
        out = self.decoder(fea_recon, points_num_list)
        coord = out.C[:, 1:] + offset
        return coord, out.F.round_()

    def decompress_partitions(self, concat_bytes: bytes):
        streams = []
        with io.BytesIO(concat_bytes) as bs:
            while bs.tell() != len(concat_bytes):
                length = int.from_bytes(bs.read(3), 'little', signed=False)
                streams.append(bs.read(length))
        # a partition's size before it is decoded: the pruning target of the finest stage in its header
        sizes = [self._parse_header(s)[1][0][0] if self.cfg.adaptive_pruning else len(s) for s in streams]
        coords, colors = [], []
        for g in self._groups(sizes):
            for c, f in self.decompress_many([streams[i] for i in g]):
                coords.append(c)
                colors.append(f)
        return torch.cat(coords, 0), torch.cat(colors, 0)

    def test_forward(self, pc_data: PCData) -> dict:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        data = self.compress(pc_data.xyz, pc_data.color)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        ME.clear_global_coordinate_manager()
        coord, color = self.decompress(data)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        ME.clear_global_coordinate_manager()
        n_org = pc_data.org_points_num[0] if pc_data.org_points_num else pc_data.xyz.shape[0]
        return {'pred': coord, 'pred_color': color, 'compressed_bytes': data, 'bpp': 8 * len(data) / n_org,
                'encode time': t1 - t0, 'decode time': t2 - t1}
