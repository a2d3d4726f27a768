from .model import PCC as Model
from .model_config import ModelConfig as Config
