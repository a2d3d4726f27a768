"""Hyper-parameters of the joint geometry + colour codec: fields and defaults of
/root/reference/models/convolutional/lossy_coord_lossy_color/model_config.py:8-45."""
from dataclasses import dataclass, fields
from typing import Tuple

from ..lossy_coord_v2.model_config import _load_with_includes


@dataclass
class ModelConfig:
    minkowski_algorithm: str = 'DEFAULT'
    conv_region_type: str = 'HYPER_CUBE'
    activation: str = 'relu'
    compressed_channels: Tuple[int, ...] = (1,)
    bottleneck_process: str = 'noise'
    bottleneck_scaler: int = 1
    bottleneck_value_bound: int = 20
    skip_encoding_fea: int = -1
    encoder_channels: Tuple[int, ...] = (8, 32)
    decoder_channels: Tuple[int, ...] = (8,)
    adaptive_pruning: bool = True
    adaptive_pruning_scaler_train: float = 1.0
    adaptive_pruning_scaler_test: float = 1.0
    geo_lossl_if_sample: Tuple[int, ...] = (1, 1)
    geo_lossl_channels: Tuple[int, ...] = (128, 128, 1)
    use_yuv_loss: bool = False
    bits_loss_factor: float = 0.2
    coord_recon_loss_factor: float = 1.0
    color_recon_loss_factor: float = 1.0
    warmup_fea_loss_steps: int = 1
    warmup_color_loss_steps: int = 1
    warmup_fea_loss_factor: float = 0.2
    warmup_color_loss_factor: float = 1.0
    linear_warmup: bool = False

    def __post_init__(self):
        for f in fields(self):
            v = getattr(self, f.name)
            if isinstance(v, list):
                setattr(self, f.name, tuple(v))
        if isinstance(self.compressed_channels, int):
            self.compressed_channels = (self.compressed_channels,)
        if isinstance(self.decoder_channels, int):
            self.decoder_channels = (self.decoder_channels,)
        if len(self.compressed_channels) == 1:
            self.compressed_channels = self.compressed_channels * len(self.geo_lossl_channels)

    @classmethod
    def from_yaml(cls, path: str) -> 'ModelConfig':
        merged = {}
        for section in _load_with_includes(path):
            merged.update(section.get('model') or {})
        return cls(**merged)


def baseline_r1() -> ModelConfig:
    """config/convolutional/lossy_coord_lossy_color/baseline_r1.yaml:2-18"""
    return ModelConfig(activation='prelu', compressed_channels=(1,), bottleneck_scaler=1, encoder_channels=(32, 64, 128),
                       decoder_channels=(64, 32), adaptive_pruning=True, geo_lossl_if_sample=(0, 1) * 5,
                       geo_lossl_channels=(128,) * 10 + (1,), use_yuv_loss=True, bits_loss_factor=0.2,
                       coord_recon_loss_factor=1.0, color_recon_loss_factor=0.02, warmup_fea_loss_steps=5000,
                       warmup_color_loss_steps=5000, warmup_fea_loss_factor=0.001, warmup_color_loss_factor=0.0001)
