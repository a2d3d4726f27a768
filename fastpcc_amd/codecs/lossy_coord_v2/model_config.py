"""Hyper-parameters of the lossy_coord_v2 codec: field names and defaults of
/root/reference/models/convolutional/lossy_coord_v2/model_config.py:8-40; `from_yaml` reads the `model:` section of the
reference's YAML files (e.g. config/convolutional/lossy_coord_v2/baseline_r1.yaml:2-13) including `# include` lines."""
import os
from dataclasses import dataclass, fields
from typing import Tuple


@dataclass
class ModelConfig:
    # network structure
    minkowski_algorithm: str = 'DEFAULT'
    conv_region_type: str = 'HYPER_CUBE'
    activation: str = 'relu'
    # compression
    compressed_channels: Tuple[int, ...] = (1,)
    bottleneck_process: str = 'noise'
    bottleneck_scaler: int = 1
    bottleneck_value_bound: int = 20
    skip_encoding_fea: int = -1
    # lossy part
    encoder_channels: Tuple[int, ...] = (4, 16, 64)
    decoder_channels: Tuple[int, ...] = (16, 4)
    adaptive_pruning: bool = True
    adaptive_pruning_scaler: float = 1.0
    # lossless part
    geo_lossl_if_sample: Tuple[int, ...] = (1, 1)
    geo_lossl_channels: Tuple[int, ...] = (128, 128, 1)
    # loss weights (training only; kept so that YAML files load)
    bits_loss_factor: float = 0.4
    coord_recon_loss_factor: float = 1.0
    warmup_fea_loss_steps: int = 1
    warmup_fea_loss_factor: float = 0.4
    linear_warmup: bool = False
    # not a key of the reference's config: True prepends one byte, the numerics version of the build that wrote the stream
    # (include/fpcc_hip.h), so that a decoder with other summation-order rules refuses the stream instead of decoding garbage.
    # Default False = the reference's byte layout exactly.
    numerics_version_in_header: bool = False

    def __post_init__(self):
        for f in fields(self):
            v = getattr(self, f.name)
            if isinstance(v, list):
                setattr(self, f.name, tuple(v))
        if isinstance(self.compressed_channels, int):
            self.compressed_channels = (self.compressed_channels,)
        if len(self.compressed_channels) == 1:
            self.compressed_channels = self.compressed_channels * len(self.geo_lossl_channels)

    @classmethod
    def from_yaml(cls, path: str) -> 'ModelConfig':
        import yaml
        merged = {}
        for section in _load_with_includes(path):
            merged.update(section.get('model') or {})
        known = {f.name for f in fields(cls)}
        unknown = set(merged) - known
        if unknown:
            raise KeyError(f'unknown model keys in {path}: {sorted(unknown)}')
        return cls(**merged)


def _load_with_includes(path: str):
    """yaml documents of `path` preceded by those of its leading `# include "<file>"` lines (lib/simple_config.py:202-204)."""
    import yaml
    out = []
    with open(path) as f:
        text = f.read()
    for line in text.splitlines():
        line = line.strip()
        if not line.startswith('# include'):
            break
        inc = line[len('# include'):].strip().strip('"\'')
        if not os.path.isabs(inc):
            inc = os.path.join(os.path.dirname(path), inc)
        out.extend(_load_with_includes(inc))
    out.append(yaml.safe_load(text) or {})
    return out


BASELINE_R1 = dict(
    activation='prelu', compressed_channels=(1,), skip_encoding_fea=1, encoder_channels=(16, 64),
    decoder_channels=(16,), adaptive_pruning=True, geo_lossl_if_sample=(0, 1) * 6,
    geo_lossl_channels=(64,) + (128,) * 11 + (1,), bits_loss_factor=0.4, warmup_fea_loss_steps=5000,
    warmup_fea_loss_factor=0.01)


def baseline_r1() -> ModelConfig:
    """config/convolutional/lossy_coord_v2/baseline_r1.yaml:2-13, the configuration BASELINE.json's metric is quoted on."""
    return ModelConfig(**BASELINE_R1)
