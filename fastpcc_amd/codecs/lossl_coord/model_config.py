"""Configuration of the float twin of the integer codec: fields and defaults of
/root/reference/models/convolutional/lossl_coord/model_config.py:8-19."""
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
    quantize_param: bool = False                    # calibrate during the test pass and write the integer parameters
    int_param_save_path: str = 'int_param.pt'
