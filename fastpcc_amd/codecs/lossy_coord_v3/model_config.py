"""Configuration of the octree codec with coded latents: fields and defaults of
/root/reference/models/convolutional/lossy_coord_v3/model_config.py:7-21."""
from dataclasses import dataclass
from typing import Tuple


@dataclass
class Config:
    channels: int = 128
    compressed_channels: int = 1
    num_latents: Tuple[int, ...] = (0, 0, 2)            # latents coded at (stride 2, stride 4, stride 8, ...)
    lossl_geo_upsample: Tuple[int, ...] = (0, 0, 0)     # (stride 2 -> 1, 4 -> 2, 8 -> 4, ...): 1 lossless, 0 top-k
    max_stride: int = 64
    torchsparse_dataflow: str = 'ImplicitGEMM'          # accepted for YAML compatibility; no meaning here

    coord_recon_loss_factor: float = 1.0
    warmup_steps: int = 0

    skip_top_scales_num: int = 0
    cal_avs_pc_evalue: bool = False

    def check(self):
        levels = int(self.max_stride).bit_length() - 1
        if 1 << levels != self.max_stride:
            raise ValueError('max_stride must be a power of two')
        if levels <= len(self.num_latents):
            raise ValueError('max_stride leaves no recurrent level above the configured ones')
        if len(self.num_latents) != len(self.lossl_geo_upsample):
            raise ValueError('num_latents and lossl_geo_upsample describe the same levels')
        first = next((i for i, v in enumerate(self.lossl_geo_upsample) if v == 1), len(self.lossl_geo_upsample))
        if not all(v == 1 for v in self.lossl_geo_upsample[first:]):
            raise ValueError('levels above the first lossless one must be lossless')
        if not all(v == 0 for v in self.num_latents[:max(first - 1, 0)]):
            raise ValueError('latents are coded from the level below the first lossless one upwards')
        return self
