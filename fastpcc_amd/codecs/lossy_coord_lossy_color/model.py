"""Joint geometry + colour codec: module tree and frame layout of `PCC` in
/root/reference/models/convolutional/lossy_coord_lossy_color/model.py:23-314 (inference).  Differences to lossy_coord_v2:
4-channel input (R, G, B in [0,1] and a constant 2), a 3-stage encoder to stride 4, a two-stage generative decoder whose
last stage also predicts the colours, and every level of the lossless coder carries residual features."""
import io
import time
from typing import List, Tuple

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

    def set_global_cm(self) -> ME.CoordinateManager:
        ME.clear_global_coordinate_manager()
        cm = ME.CoordinateManager(D=3)
        ME.set_global_coordinate_manager(cm)
        return cm

    def get_sparse_pc(self, xyz: torch.Tensor, color: torch.Tensor) -> ME.SparseTensor:
        cm = self.set_global_cm()
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
        with io.BytesIO() as bs:
            for v in coord_offset.tolist():
                bs.write(int(v).to_bytes(2, 'little', signed=False))
            if self.cfg.adaptive_pruning:
                for counts in points_num_list:
                    bs.write(int(counts[0]).to_bytes(3, 'little', signed=False))
            bs.write(em_bytes)
            return bs.getvalue()

    def compress_partitions(self, batched_coord: List[torch.Tensor], batched_color: List[torch.Tensor]) -> bytes:
        parts = [self.compress(c, f) for c, f in zip(batched_coord[1:], batched_color[1:])]
        return b''.join(len(s).to_bytes(3, 'little', signed=False) + s for s in parts)

    @hipops.no_gc_pause
    @torch.no_grad()
    def decompress(self, compressed_bytes: bytes) -> Tuple[torch.Tensor, torch.Tensor]:
        dev = next(self.parameters()).device
        with io.BytesIO(compressed_bytes) as bs:
            coord_offset = [int.from_bytes(bs.read(2), 'little', signed=False) for _ in range(3)]
            points_num_list = None
            if self.cfg.adaptive_pruning:
                points_num_list = [[int.from_bytes(bs.read(3), 'little', signed=False)]
                                   for _ in range(len(self.cfg.decoder_channels))]
            em_bytes = bs.read()
        offset = torch.tensor(coord_offset, dtype=torch.int32, device=dev)      # before anything is queued (pageable H2D waits)
        fea_recon = self.em_lossless_based.decompress(em_bytes, self.set_global_cm())
        out = self.decoder(fea_recon, points_num_list)
        coord = out.C[:, 1:] + offset
        return coord, out.F.round_()

    def decompress_partitions(self, concat_bytes: bytes):
        coords, colors = [], []
        with io.BytesIO(concat_bytes) as bs:
            while bs.tell() != len(concat_bytes):
                length = int.from_bytes(bs.read(3), 'little', signed=False)
                c, f = self.decompress(bs.read(length))
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
