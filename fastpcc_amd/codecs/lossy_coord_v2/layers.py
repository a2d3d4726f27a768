"""Network pieces of the lossy_coord_v2 codec (inference paths).

Same class names, constructor arguments and sub-module attribute names as
/root/reference/models/convolutional/lossy_coord_v2/layers.py:28-415, so a reference checkpoint's state_dict keys load
unchanged; the forward passes are written against fastpcc_amd.engine (fused conv+bias+activation(+clip) launches, lazy
channel concatenation, fused top-k pruning).  Training-time members of the reference (`train_forward`, `get_target`,
`get_coord_recon_loss`, `BoundFunction.backward`) are outside this inference path.
"""
from typing import List, Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import engine as ME
from ... import hipops as ops
from ...sparse_conv_layers import ConvBlock, ConvTransBlock, GenConvTransBlock, MEMLPBlock, \
    NNSequentialWithConvBlockArgs, NNSequentialWithConvTransBlockArgs, mlp_chain_forward


def _run(seq: nn.Sequential, x, last_clip: float = 0.0):
    """nn.Sequential forward that lets the LAST block fuse a clamp into its epilogue."""
    mods = list(seq)
    for m in mods[:-1]:
        x = m(x)
    return mods[-1](x, clip=last_clip) if last_clip > 0 else mods[-1](x)


def classify_head(seq: nn.Sequential, fea):
    """the decoder's classify block -- ConvBlock(ch, ch // 2, 1, 1) + ConvBlock(ch // 2, 1, 1, 1), layers.py:105-110 -- as ONE launch
    when it is the narrow per-point head fpcc_pointwise_head_f32 takes (16 -> 8 -> 1): evaluated separately the hidden layer is
    zero-padded to 32 MFMA columns on the 8 N candidates.  Same bits (the hidden layer keeps the summation order its separate
    evaluation would use at this row count); falls back to the two blocks otherwise."""
    blocks = list(seq)
    ok = len(blocks) == 2 and not torch.is_grad_enabled() and len(fea.parts) == 1
    if ok:
        a, b = blocks
        ok = all(isinstance(m, ConvBlock) and m.bn is None and m.conv.ks == 1 and
                 (m.act_module is None or isinstance(m.act_module, (ME.MinkowskiPReLU, ME.MinkowskiReLU))) for m in blocks) and \
            b.conv.out_channels == 1 and a.conv.out_channels == b.conv.in_channels and \
            ops.pointwise_head_ok(a.conv.in_channels, a.conv.out_channels)
    x = fea.parts[0] if ok else None
    if not ok or x.dtype != torch.float32 or x.stride(1) != 1 or (x.shape[0] > 1 and x.stride(0) % 4) or x.data_ptr() % 16:
        return seq(fea)
    da, db = a.conv._derived(), b.conv._derived()
    act1, act2 = ME._act_of(a.act_module), ME._act_of(b.act_module)
    cm = fea.coordinate_manager
    orders = {ME.summation_order('k1', a.conv.in_channels, 0, a.conv.out_channels, rows)
              for rows in cm.cloud_rows(cm._map(fea.coordinate_map_key)) if rows > 0}
    if len(orders) != 1 or not orders <= {0, 1}:     # (independent clouds on either side of PAD_MIN_ROWS: the blocks sort that out)
        return seq(fea)
    order1 = orders.pop()
    out = ops.pointwise_head(x, da['w'].reshape(a.conv.in_channels, a.conv.out_channels), da['b'], act1.kind, act1.slope, order1,
                             db['w'].reshape(-1), db['b'], act2.kind, act2.slope)
    return ME.SparseTensor(out, coordinate_map_key=fea.coordinate_map_key, coordinate_manager=fea.coordinate_manager)


class Encoder(nn.Module):
    """conv3(in->c0) @stride 1, then per extra channel entry: conv2s2 + conv3 (layers.py:28-72)."""

    def __init__(self, in_channels: int, intra_channels: Tuple[int, ...], requires_points_num_list: bool,
                 points_num_scaler: float, region_type: str, act: Optional[str]):
        super().__init__()
        self.requires_points_num_list = requires_points_num_list
        self.points_num_scaler = points_num_scaler
        stages = [ConvBlock(in_channels, intra_channels[0], 3, 1, region_type=region_type, act=act)]
        for c_prev, c in zip(intra_channels[:-1], intra_channels[1:]):
            stages.append(nn.Sequential(
                ConvBlock(c_prev, c, 2, 2, region_type='HYPER_CUBE', act=act),
                ConvBlock(c, c, 3, 1, region_type=region_type, act=act)))
        self.blocks = nn.ModuleList(stages)

    def forward(self, x) -> Tuple[ME.SparseTensor, Optional[List[List[int]]]]:
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
        return x, [[int(n * self.points_num_scaler) for n in c] for c in counts]


class Decoder(nn.Module):
    """Generative upsampling with adaptive top-k pruning back to stride 1 (layers.py:75-180)."""

    def __init__(self, in_channels: int, intra_channels: Tuple[int, ...], region_type: str, act: Optional[str]):
        super().__init__()
        self.upsample_blocks = nn.ModuleList()
        self.classify_blocks = nn.ModuleList()
        prev = in_channels
        n_stage = len(intra_channels)
        for i, ch in enumerate(intra_channels):
            up = nn.Sequential()
            if i == n_stage - 1:
                up.append(ConvBlock(prev, prev, 3, 1, region_type=region_type, act=act))
            up.append(GenConvTransBlock(prev, ch, 2, 2, region_type='HYPER_CUBE', act=act))
            if i != n_stage - 1:
                up.append(ConvBlock(ch, ch, 3, 1, region_type=region_type, act=act))
            self.upsample_blocks.append(up)
            self.classify_blocks.append(nn.Sequential(ConvBlock(ch, ch // 2, 1, 1, act=act),
                                                      ConvBlock(ch // 2, 1, 1, 1, act=None)))
            prev = ch
        self.pruning = ME.MinkowskiPruning()

    def forward(self, fea, points_num_list, coord_offset=None):
        if self.training:
            return self.train_forward(fea, points_num_list, coord_offset)      # third argument: the target key
        return self.test_forward(fea, points_num_list, coord_offset)

    def train_forward(self, fea, points_num_list, target_key: ME.CoordinateMapKey) -> dict:
        """per upsampling stage: binary cross-entropy of the occupancy logits against the true finer coordinates, then
        prune to (adaptive top-k | true) candidates (layers.py:118-137)"""
        loss = {}
        n_stage = len(self.upsample_blocks)
        inv = [1 / sum(c) for c in points_num_list] if points_num_list is not None else None
        for i, (up, classify) in enumerate(zip(self.upsample_blocks, self.classify_blocks)):
            fea = up(fea)
            pred = classify(fea)
            keep = self.get_keep_train(pred, points_num_list)
            target = self.get_target(pred, target_key)
            keep |= target
            loss[f'coord_{n_stage - i - 1}_recon_loss'] = F.binary_cross_entropy_with_logits(
                pred.F.squeeze(1), target.to(pred.F.dtype), reduction='sum')
            if i != n_stage - 1:
                fea = self.pruning(fea, keep.to(torch.uint8))
        if inv is not None and len(inv) != 1:
            total = sum(inv)
            for i in range(n_stage):
                loss[f'coord_{i}_recon_loss'] = loss[f'coord_{i}_recon_loss'] * (inv[i] / total * len(inv))
        return loss

    @torch.no_grad()
    def get_target(self, pred: ME.SparseTensor, target_key: ME.CoordinateMapKey) -> torch.Tensor:
        """bool [n]: which generated candidates are voxels of the (strided) target set (layers.py:182-190)"""
        cm = pred.coordinate_manager
        gen = cm._map(pred.coordinate_map_key)
        tgt = cm._map(cm.stride(target_key, pred.tensor_stride))
        if not gen.generated or tgt.parent is not gen.parent:
            raise NotImplementedError('the target set must be a child map of the map the candidates were generated from')
        return ops.child_mask(tgt.child_row).bool()

    @torch.no_grad()
    def get_keep_train(self, pred: ME.SparseTensor, points_num_list: Optional[List[List[int]]]) -> torch.Tensor:
        """training-time variant of get_keep for a batch: per sample, the logits above its own k-th value, or the maximum
        of their 2x2x2 cell.  Bookkeeping under no_grad, written with tensor ops (one stage of upsampling)."""
        cm = pred.coordinate_manager
        gen = cm._map(pred.coordinate_map_key)
        if not gen.generated or len(self.upsample_blocks) != 1:
            raise NotImplementedError('training-time pruning supports one generative stage (decoder_channels of length 1)')
        logits = pred.F.view(-1, 8)
        if points_num_list is None:
            return ((logits > 0) | (logits == logits.max(1, keepdim=True).values)).view(-1)
        targets = points_num_list.pop()
        edges = cm.batch_offsets(gen.parent)
        keep = torch.empty(logits.numel(), dtype=torch.uint8, device=logits.device)
        for tgt, a, b in zip(targets, edges[:-1], edges[1:]):
            if not (b - a) * 8 > tgt:
                raise ValueError('fewer candidates than points to keep')
            keep[8 * a: 8 * b] = ops.topk_keep(logits[a:b].reshape(-1), int(tgt))      # the sample's own k-th value
        return keep.bool()

    @torch.no_grad()
    def test_forward(self, fea, points_num_list, coord_offset: Optional[torch.Tensor] = None):
        """Returns xyz int32 [N, 3] (plus coord_offset, a device int32[3], when given).  On a batch of independent clouds
        (ME.CoordinateManager(clouds=B); coord_offset int32 [B, 3]; points_num_list entries hold one target per cloud): the list of
        the clouds' xyz tensors."""
        last = len(self.upsample_blocks) - 1
        cm = fea.coordinate_manager
        top = cm._map(fea.coordinate_map_key)      # local maxima are taken inside the voxels of this level
        for i, (up, classify) in enumerate(zip(self.upsample_blocks, self.classify_blocks)):
            fea = up(fea)
            keep = self.get_keep(classify_head(classify, fea), points_num_list, top)
            if i != last:
                fea = self.pruning(fea, keep)
            else:
                gen = cm._map(fea.coordinate_map_key)
                if not gen.generated:
                    raise RuntimeError('decoder output is expected on a generated coordinate set')
                if not cm.independent_clouds:
                    xyz, count = ops.compact_coords(gen.parent.keys, keep, gen.level, gen.bits, coord_offset)
                    return xyz[:int(count.item())]
                edges = cm.batch_offsets(gen.parent)
                parts = [ops.compact_coords(gen.parent.keys[a:b], keep[8 * a: 8 * b], gen.level, gen.bits,
                                            None if coord_offset is None else coord_offset[c].contiguous())
                         for c, (a, b) in enumerate(zip(edges[:-1], edges[1:]))]
                counts = torch.cat([count for _, count in parts]).tolist()              # one read-back for all clouds
                return [xyz[:n] for (xyz, _), n in zip(parts, counts)]

    @torch.no_grad()
    def get_keep(self, pred: ME.SparseTensor, points_num_list: Optional[List[List[int]]], top=None) -> torch.Tensor:
        """uint8 [n]: logit above the adaptive threshold, or the maximum of its cell (layers.py:151-180).  A cell is a voxel
        of the decoder's INPUT level (`top`, tensor stride 2^stages: max_stride_lossy_recon in the reference) -- the 8
        siblings in the first stage, 64 candidates in the second, 512 in the third.  The threshold is a cloud's own: on a batch of
        independent clouds every cloud's candidates are ranked among themselves."""
        cm = pred.coordinate_manager
        gen = cm._map(pred.coordinate_map_key)
        if not gen.generated:
            raise NotImplementedError('get_keep expects the candidates of a generative upsampling')
        logits = pred.F.view(-1)
        parent = gen.parent
        cell = None
        if top is not None and parent is not top:
            # follow the parent links of the candidates' parents up to the decoder's input level
            cell, m = parent.parent_of, parent.parent
            while m is not top:
                if m is None or m.parent_of is None:
                    raise RuntimeError('candidate set is not below the decoder input level')
                cell, m = m.parent_of[cell.long()], m.parent
            cell = cell.to(torch.int32).contiguous()
        if points_num_list is None:
            # adaptive_pruning = False (layers.py:176-180): fixed threshold 0, plus the maximum of every cell
            if cell is None:
                cells = pred.F.view(-1, 8)
                return ((cells > 0) | (cells == cells.max(1, keepdim=True).values)).view(-1).to(torch.uint8)
            per_group = pred.F.view(-1, 8).max(1).values
            cell_max = torch.full((top.n,), float('-inf'), dtype=per_group.dtype, device=per_group.device)
            cell_max.scatter_reduce_(0, cell.long(), per_group, reduce='amax', include_self=True)
            cells = pred.F.view(-1, 8)
            return ((cells > 0) | (cells == cell_max[cell.long()][:, None])).view(-1).to(torch.uint8)
        return adaptive_keep(cm, parent, top, logits, cell, points_num_list.pop())


def adaptive_keep(cm, parent, top, logits: torch.Tensor, cell: Optional[torch.Tensor], target: List[int]) -> torch.Tensor:
    """uint8 [8 * parent.n]: the candidates (8 per row of `parent`) above the threshold that keeps `target` of them, or the maximum of
    their cell (`cell`: the row of `top` every parent row lies in; None = the 8 siblings).  One target, one ranking per cloud: on a
    batch of independent clouds (ME.CoordinateManager(clouds=B)) every cloud's candidates are ranked among themselves."""
    if cm.independent_clouds:
        edges = cm.batch_offsets(parent)
        if len(target) != len(edges) - 1:
            raise ValueError('one pruning target per cloud expected')
        top_edges = cm.batch_offsets(top) if cell is not None else None
        keep = []
        for c, (a, b) in enumerate(zip(edges[:-1], edges[1:])):
            if not 8 * (b - a) > target[c]:
                raise ValueError('fewer candidates than points to keep')
            if cell is None:
                keep.append(ops.topk_keep(logits[8 * a: 8 * b], target[c]))
            else:
                keep.append(ops.topk_keep_cells(logits[8 * a: 8 * b], (cell[a:b] - top_edges[c]).contiguous(),
                                                top_edges[c + 1] - top_edges[c], target[c]))
        return torch.cat(keep)
    if len(target) != 1:
        raise NotImplementedError('batch size 1 at test time, as in the reference (model.py:121)')
    if not logits.numel() > target[0]:
        raise ValueError('fewer candidates than points to keep')
    if cell is None:
        return ops.topk_keep(logits, target[0])
    return ops.topk_keep_cells(logits, cell, top.n, target[0])


class HyperDecoderUpsample(nn.Module):
    """Per level: (transposed conv k2s2 onto a given key | conv3) + conv3 (layers.py:201-227)."""

    def __init__(self, in_channels: Tuple[int, ...], out_channels: Tuple[int, ...], if_sample: Tuple[int, ...],
                 region_type: str, act: Optional[str]):
        super().__init__()
        self.blocks = nn.ModuleList()
        for c_in, c_out, up in zip(in_channels, out_channels, if_sample):
            if up:
                head = ConvTransBlock(max(c_in, 1), c_out, 2, 2, region_type=region_type, act=act)
                seq = NNSequentialWithConvTransBlockArgs
            else:
                head = ConvBlock(max(c_in, 1), c_out, 3, 1, region_type=region_type, act=act)
                seq = NNSequentialWithConvBlockArgs
            self.blocks.append(seq(head, ConvBlock(c_out, c_out, 3, 1, region_type=region_type, act=act)))

    def __len__(self):
        return len(self.blocks)

    def __getitem__(self, idx):
        return self.blocks[idx]


class HyperDecoderGenUpsample(nn.Module):
    """Per upsampling level: generative conv k2s2 (C -> C/4) + conv3 (-> 1 occupancy logit) (layers.py:230-249)."""

    def __init__(self, in_channels: Tuple[int, ...], if_sample: Tuple[int, ...], region_type: str, act: Optional[str]):
        super().__init__()
        self.blocks = nn.ModuleList()
        for c_in, up in zip(in_channels, if_sample):
            if not up:
                self.blocks.append(None)
                continue
            mid = max(c_in // 4, 1)
            self.blocks.append(nn.Sequential(
                GenConvTransBlock(c_in, mid, 2, 2, region_type=region_type, act=act),
                ConvBlock(mid, 1, 3, 1, region_type=region_type, act=None)))

    def __getitem__(self, idx):
        return self.blocks[idx]


class SubResidualGeoLossl(nn.Module):
    """cat(fea, prediction) -> conv3 -> conv3 -> clamp to +-bound (layers.py:252-271)."""

    def __init__(self, in_ch, out_ch, region_type, act, bottleneck_value_bound: int):
        super().__init__()
        self.blocks = nn.Sequential(
            ConvBlock(in_ch + in_ch, in_ch, 3, 1, region_type=region_type, act=act),
            ConvBlock(in_ch, out_ch, 3, 1, region_type=region_type, act=None))
        self.register_buffer('bound', torch.tensor(bottleneck_value_bound), persistent=False)
        self._bound = float(bottleneck_value_bound)

    def forward(self, x, y):
        return _run(self.blocks, ME.cat(x, y), last_clip=self._bound)


class ResidualGeoLossl(nn.Module):
    def __init__(self, in_channels: Tuple[int, ...], out_channels: Tuple[int, ...], region_type: str,
                 act: Optional[str], bottleneck_value_bound: int, skip_encoding_fea: int):
        super().__init__()
        self.blocks = nn.ModuleList()
        for idx, (c_in, c_out) in enumerate(zip(in_channels, out_channels)):
            self.blocks.append(SubResidualGeoLossl(c_in, c_out, region_type, act, bottleneck_value_bound)
                               if idx > skip_encoding_fea else None)

    def __len__(self):
        return len(self.blocks)

    def __getitem__(self, idx):
        return self.blocks[idx]


class SubDecoderGeoLossl(nn.Module):
    """MLP(residual) ++ prediction -> MLP (layers.py:294-315)."""

    def __init__(self, in_ch, in_ch2, out_ch, region_type, act):
        super().__init__()
        self.residual_decoder = nn.Sequential(MEMLPBlock(in_ch, out_ch // 2, act=act),
                                              MEMLPBlock(out_ch // 2, out_ch, act=act))
        self.decoder = nn.Sequential(MEMLPBlock(out_ch + in_ch2, out_ch, act=act),
                                     MEMLPBlock(out_ch, out_ch, act=act))

    def forward(self, x, y: ME.SparseTensor):
        # inference: the four per-point layers and the concatenation as one launch (activations stay in LDS; same bits)
        fused = mlp_chain_forward([*self.residual_decoder, *self.decoder], x, y, cat_layer=len(self.residual_decoder))
        if fused is not None:
            return ME.SparseTensor(fused, coordinate_map_key=y.coordinate_map_key, coordinate_manager=y.coordinate_manager)
        if isinstance(x, torch.Tensor):
            x = ME.SparseTensor(x, coordinate_map_key=y.coordinate_map_key, coordinate_manager=y.coordinate_manager)
        return self.decoder(ME.cat(self.residual_decoder(x), y))


class SubDecoderGeoLossl2(nn.Module):
    def __init__(self, in_ch, in_ch2, out_ch, region_type, act):
        super().__init__()
        self.decoder = nn.Sequential(MEMLPBlock(in_ch2, out_ch, act=act), MEMLPBlock(out_ch, out_ch, act=act))

    def forward(self, x):
        fused = mlp_chain_forward(list(self.decoder), x)
        if fused is not None:
            return ME.SparseTensor(fused, coordinate_map_key=x.coordinate_map_key, coordinate_manager=x.coordinate_manager)
        return self.decoder(x)


class DecoderGeoLossl(nn.Module):
    def __init__(self, in_channels: Tuple[int, ...], in_channels2: Tuple[int, ...], out_channels: Tuple[int, ...],
                 region_type: str, act: Optional[str], skip_encoding_fea: int):
        super().__init__()
        self.blocks = nn.ModuleList()
        for idx, (c_in, c_out, c_in2) in enumerate(zip(in_channels, out_channels, in_channels2)):
            cls = SubDecoderGeoLossl if idx > skip_encoding_fea else SubDecoderGeoLossl2
            self.blocks.append(cls(c_in, c_in2, c_out, region_type, act))

    def __len__(self):
        return len(self.blocks)

    def __getitem__(self, idx):
        return self.blocks[idx]


class EncoderGeoLossl(nn.Module):
    """The analysis pyramid of the lossless coder: per level (conv k2s2 | conv3) + conv3, with an MLP head on the levels
    whose features are coded; the last head is clamped to +-bound (layers.py:357-415)."""

    def __init__(self, in_channels: Tuple[int, ...], out_channels: Tuple[int, ...], if_sample: Tuple[int, ...],
                 region_type: str, act: Optional[str], bottleneck_value_bound: int, skip_encoding_fea: int):
        super().__init__()
        if len(in_channels) + 1 != len(out_channels):
            raise ValueError('out_channels must have one more entry than in_channels')
        self.blocks_out_first = MEMLPBlock(in_channels[0], out_channels[0], act=act) if skip_encoding_fea < 0 else None
        self.blocks = nn.ModuleList()
        self.blocks_out = nn.ModuleList()
        for idx, (c_in, c_out, down) in enumerate(zip(in_channels, out_channels[1:], if_sample)):
            wide = max(c_in, c_out)
            first = ConvBlock(c_in, c_in, 2, 2, region_type=region_type, act=act) if down else \
                ConvBlock(c_in, c_in, 3, 1, region_type=region_type, act=act)
            self.blocks.append(nn.Sequential(first, ConvBlock(c_in, wide, 3, 1, region_type=region_type, act=act)))
            self.blocks_out.append(MEMLPBlock(wide, c_out, act=act) if idx >= skip_encoding_fea else None)
        self.out_channels = out_channels
        self.register_buffer('bound', torch.tensor(bottleneck_value_bound), persistent=False)
        self._bound = float(bottleneck_value_bound)

    def __len__(self):
        return len(self.blocks)

    def forward(self, x: ME.SparseTensor, batch_size: int) -> List[ME.SparseTensor]:
        if batch_size != 1 and not self.training:
            raise NotImplementedError('batch size 1 at test time')
        outs = [self.blocks_out_first(x) if self.blocks_out_first is not None else x]
        last = len(self.blocks) - 1
        for i, (block, head) in enumerate(zip(self.blocks, self.blocks_out)):
            x = block(x)
            if head is None:
                outs.append(x)
            else:
                outs.append(head(x, clip=self._bound) if i == last else head(x))
        if self.blocks_out[last] is None:
            raise NotImplementedError('the bottom level must carry a coded feature')
        return outs
