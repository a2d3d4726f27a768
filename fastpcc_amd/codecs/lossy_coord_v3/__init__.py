from .model import Model
from .model_config import Config
