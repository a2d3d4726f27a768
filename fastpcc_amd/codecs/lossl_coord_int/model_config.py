"""Configuration of the integer codec: fields and defaults of
/root/reference/models/convolutional/lossl_coord_int/model_config.py:8-17 (`skip_top_scales_num` = rate point - 1, the
r1..r6 YAMLs under config/convolutional/lossl_coord/kitti_ford_test_int_r*.yaml)."""
from dataclasses import dataclass


@dataclass
class Config:
    torchsparse_dataflow: str = 'ImplicitGEMM'      # accepted for YAML compatibility; no meaning here
    channels: int = 256
    max_stride_wo_recurrent: int = 2048
    max_stride: int = 8192
    fea_stride: int = 16
    use_more_ch_for_multi_step_pred: bool = False
    skip_top_scales_num: int = 0
    cal_avs_pc_evalue: bool = False
