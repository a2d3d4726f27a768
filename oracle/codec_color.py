"""TEST ORACLE -- not product code.

CPU restatement of the joint geometry + colour codec (/root/reference/models/convolutional/lossy_coord_lossy_color/
model.py:144-298, layers.py:30-233) on top of oracle/codec_v2.py, whose lossless entropy model, layers and framing it
shares (the reference shares them too: geo_lossl_em.py is imported by both models).  Parity: as oracle/codec_v2.py."""
import io
from typing import List

import numpy as np
import torch

from . import coords as oc
from .codec_v2 import Feature, OracleV2, _r, _u


class OracleColor(OracleV2):
    def encoder(self, x: Feature):
        counts = []
        n_blocks = len(self.cfg.encoder_channels)
        x = self.conv_block('encoder.blocks.0', x, 'k3')
        if n_blocks > 1:
            counts.append([x.level.n])
        for i in range(1, n_blocks):
            x = self.conv_block(f'encoder.blocks.{i}.0', x, 'k2s2')
            x = self.conv_block(f'encoder.blocks.{i}.1', x, 'k3')
            if i != n_blocks - 1:
                counts.append([x.level.n])
        s = self.cfg.adaptive_pruning_scaler_test
        return x, [[int(n * s) for n in c] for c in counts]

    def get_keep_cells(self, pred: Feature, top: oc.Level, target: int) -> np.ndarray:
        """Decoder.get_keep (layers.py:183-209): max-pool with stride 2^stages / pred.stride onto the coarsest decoder
        level, un-pool, k-th value threshold"""
        v = pred.f.reshape(-1).numpy()
        q = pred.level.coords.copy()
        q[:, 1:] = q[:, 1:] // top.stride * top.stride
        cell = top.rows_of(q)
        cell_max = np.full(top.n, -np.inf, dtype=np.float32)
        np.maximum.at(cell_max, cell, v)
        not_max = (v - cell_max[cell]) != 0
        ranked = np.sort(v[not_max])
        kth = v.shape[0] - target
        assert v.shape[0] > target and 1 <= kth <= ranked.shape[0]
        return (v > ranked[kth - 1]) | ~not_max

    def decoder(self, fea: Feature, points_num_list: List[List[int]]):
        top = fea.level
        n_stage = len(self.cfg.decoder_channels)
        keep = None
        for i in range(n_stage):
            fea = self.conv_block(f'decoder.upsample_blocks.{i}.0', fea, 'gen')
            fea = self.conv_block(f'decoder.upsample_blocks.{i}.1', fea, 'k3')
            pred = self.conv_block(f'decoder.classify_blocks.{i}.0', fea, 'k3')
            pred = self.conv_block(f'decoder.classify_blocks.{i}.1', pred, 'k3')
            keep = self.get_keep_cells(pred, top, points_num_list.pop()[0])
            if i != n_stage - 1:
                fea = Feature(fea.f[torch.from_numpy(keep)], oc.Level(fea.level.coords[keep], fea.level.stride))
        flags = Feature(torch.from_numpy(keep.astype(np.float32))[:, None].expand(-1, 2).contiguous(), fea.level)
        x = self.conv_block('decoder.predict_block.0', fea, 'k3', x2=flags)
        x = self.conv_block('decoder.predict_block.1', x, 'k3')
        x = self.conv_block('decoder.predict_block.2', x, 'k3')
        rgb = (x.f[torch.from_numpy(keep)].clip(0, 1) * 255)
        return fea.level.coords[keep][:, 1:], rgb

    def compress(self, batched_coord: np.ndarray, color: np.ndarray) -> bytes:
        c = np.asarray(batched_coord, dtype=np.int64)
        offset = c[:, 1:].min(0)
        c = c.copy()
        c[:, 1:] -= offset
        level = oc.Level(c, 1)
        assert level.n == len(c), 'duplicate voxels'
        col = torch.from_numpy(np.asarray(color)[level.order].astype(np.float32))
        feats = torch.cat((col / 255, torch.full((level.n, 1), 2.0)), 1)
        fea, counts = self.encoder(Feature(feats, level))
        em = self.em_compress(fea)
        out = b''.join(_u(v, 2) for v in offset.tolist())
        if self.cfg.adaptive_pruning:
            out += b''.join(_u(cnt[0], 3) for cnt in counts)
        return out + em

    def decompress(self, data: bytes):
        with io.BytesIO(data) as bs:
            offset = [_r(bs, 2) for _ in range(3)]
            counts = [[_r(bs, 3)] for _ in range(len(self.cfg.decoder_channels))] if self.cfg.adaptive_pruning else None
            em = bs.read()
        xyz, rgb = self.decoder(self.em_decompress(em), counts)
        return (xyz + np.array(offset, dtype=np.int64)).astype(np.int32), torch.round(rgb).numpy()
