"""Encoder / decoder of the joint geometry + colour codec (inference paths), module tree of
/root/reference/models/convolutional/lossy_coord_lossy_color/layers.py:30-233.  The hierarchical lossless part is shared
with lossy_coord_v2 (the reference keeps two identical copies of those classes, layers.py:336-550)."""
from typing import List, Optional, Tuple

import torch
import torch.nn as nn

from ... import engine as ME
from ... import hipops as ops
from ...sparse_conv_layers import ConvBlock, GenConvTransBlock
from ..lossy_coord_v2.layers import DecoderGeoLossl, EncoderGeoLossl, HyperDecoderGenUpsample, HyperDecoderUpsample, \
    ResidualGeoLossl, adaptive_keep  # noqa: F401  (re-exported)


class Encoder(nn.Module):
    def __init__(self, in_channels: int, out_channels: int, intra_channels: Tuple[int, ...], requires_points_num_list: bool,
                 points_num_scaler_train: float, points_num_scaler_test: float, region_type: str, act: Optional[str]):
        super().__init__()
        self.requires_points_num_list = requires_points_num_list
        self.points_num_scaler_train = points_num_scaler_train
        self.points_num_scaler_test = points_num_scaler_test
        stages = [ConvBlock(in_channels, intra_channels[0], 3, 1, region_type=region_type, act=act)]
        prev = intra_channels[0]
        tail = intra_channels[1:]
        for i, ch in enumerate(tail):
            stages.append(nn.Sequential(
                ConvBlock(prev, ch, 2, 2, region_type='HYPER_CUBE', act=act),
                ConvBlock(ch, out_channels if i == len(tail) - 1 else ch, 3, 1, region_type=region_type, act=act)))
            prev = ch
        self.blocks = nn.ModuleList(stages)

    def forward(self, x):
        counts = []
        last = len(self.blocks) - 1
        for i, block in enumerate(self.blocks):
            x = block(x)
            if i != last:
                cm = x.coordinate_manager
                edges = cm.batch_offsets(cm._map(x.coordinate_map_key))
                counts.append([b - a for a, b in zip(edges[:-1], edges[1:])])
        if not self.requires_points_num_list:
            return x, None
        scaler = self.points_num_scaler_train if self.training else self.points_num_scaler_test
        return x, [[int(n * scaler) for n in c] for c in counts]


class Decoder(nn.Module):
    def __init__(self, in_channels: int, out_channels: int, intra_channels: Tuple[int, ...], region_type: str,
                 act: Optional[str], use_yuv_loss: bool):
        super().__init__()
        self.use_yuv_loss = use_yuv_loss
        self.upsample_blocks = nn.ModuleList()
        self.classify_blocks = nn.ModuleList()
        prev = in_channels
        for ch in intra_channels:
            self.upsample_blocks.append(nn.Sequential(
                GenConvTransBlock(prev, ch, 2, 2, region_type='HYPER_CUBE', act=act),
                ConvBlock(ch, ch, 3, 1, region_type=region_type, act=act)))
            self.classify_blocks.append(nn.Sequential(
                ConvBlock(ch, ch, 3, 1, region_type=region_type, act=act),
                ConvBlock(ch, 1, 3, 1, region_type=region_type, act=None)))
            prev = ch
        self.predict_block = nn.Sequential(
            ConvBlock(prev + 2, prev // 2, 3, 1, region_type=region_type, act=act),
            ConvBlock(prev // 2, prev // 2, 3, 1, region_type=region_type, act=act),
            ConvBlock(prev // 2, out_channels, 3, 1, region_type=region_type, act=None))
        self.pruning = ME.MinkowskiPruning()

    def forward(self, fea, points_num_list):
        if self.training:
            raise NotImplementedError('training path is not part of this build')
        return self.test_forward(fea, points_num_list)

    @torch.no_grad()
    def test_forward(self, fea, points_num_list) -> ME.SparseTensor:
        n_stage = len(self.upsample_blocks)
        cm = fea.coordinate_manager
        top = cm._map(fea.coordinate_map_key)              # the coarsest decoder level: cells of the local-maximum rule
        keep = None
        for i, (up, classify) in enumerate(zip(self.upsample_blocks, self.classify_blocks)):
            fea = up(fea)
            keep = self.get_keep(classify(fea), points_num_list, top)
            if i != n_stage - 1:
                fea = self.pruning(fea, keep)
        flags = keep.to(torch.float32)[:, None].expand(-1, 2).contiguous()
        fea = ME.cat(fea, ME.SparseTensor(flags, coordinate_map_key=fea.coordinate_map_key, coordinate_manager=cm))
        out = self.pruning(self.predict_block(fea), keep)
        rgb = out.F.clip_(0, 1).mul_(255)                   # inverse_transform_for_color, eval branch (layers.py:231-233)
        return ME.SparseTensor(rgb, coordinate_map_key=out.coordinate_map_key, coordinate_manager=cm)

    @torch.no_grad()
    def get_keep(self, pred: ME.SparseTensor, points_num_list: Optional[List[List[int]]], top) -> torch.Tensor:
        cm = pred.coordinate_manager
        gen = cm._map(pred.coordinate_map_key)
        if not gen.generated:
            raise NotImplementedError('get_keep expects the candidates of a generative upsampling')
        if points_num_list is None:
            raise NotImplementedError('adaptive_pruning=False is not part of the in-scope configurations')
        logits = pred.F.reshape(-1).contiguous()
        parent = gen.parent
        cell = None
        if parent is not top:
            # cells are the voxels of `top`: follow the parent links of the candidates' parents up to that level
            cell, m = parent.parent_of, parent.parent
            while m is not top:
                if m is None or m.parent_of is None:
                    raise RuntimeError('candidate set is not below the coarsest decoder level')
                cell, m = m.parent_of[cell.long()], m.parent
            cell = cell.to(torch.int32).contiguous()
        return adaptive_keep(cm, parent, top, logits, cell, points_num_list.pop())
