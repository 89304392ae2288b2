from .config import ImageEncoderCfg, ModelCfg, TextDecoderCfg, get_model_config, list_models
from .cruller import Cruller
