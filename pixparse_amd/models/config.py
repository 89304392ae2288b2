"""Model config dataclasses + JSON registry (ref: models/config.py:15-67), without simple_parsing."""
import copy
import json
from dataclasses import dataclass, field, fields, is_dataclass
from pathlib import Path
from typing import Optional, Tuple

from ..utils.name_utils import _natural_key, clean_name

_MODEL_CONFIG_PATHS = [Path(__file__).parent / 'configs']
_MODEL_CONFIGS = {}


class _Serializable:
    """the two simple_parsing.Serializable methods the reference relies on (load / to_dict)."""

    @classmethod
    def from_dict(cls, d: dict):
        kw = {}
        for f in fields(cls):
            if f.name not in d:
                continue
            v = d[f.name]
            sub = _SUBTYPES.get((cls.__name__, f.name))
            if sub is not None and isinstance(v, dict):
                v = sub.from_dict(v)
            elif isinstance(v, list):
                v = tuple(v)
            kw[f.name] = v
        return cls(**kw)

    @classmethod
    def load(cls, path):
        with open(path) as f:
            return cls.from_dict(json.load(f))

    def to_dict(self):
        out = {}
        for f in fields(self):
            v = getattr(self, f.name)
            out[f.name] = v.to_dict() if is_dataclass(v) else (list(v) if isinstance(v, tuple) else v)
        return out


@dataclass
class ImageEncoderCfg(_Serializable):
    name: str = 'vit_base_patch16_224'
    image_fmt: str = 'L'
    image_size: Optional[Tuple[int, int]] = (576, 448)
    pretrained: bool = True
    pretrained_path: Optional[str] = None   # local timm state dict (file or directory); see models/pretrained.py


@dataclass
class TextDecoderCfg(_Serializable):
    name: str = 'facebook/bart-base'
    pretrained: bool = True
    num_decoder_layers: Optional[int] = 4
    max_length: Optional[int] = 1024
    pad_token_id: Optional[int] = None
    pretrained_path: Optional[str] = None   # local HF BART state dict (file or directory); see models/pretrained.py


@dataclass
class ModelCfg(_Serializable):
    image_encoder: ImageEncoderCfg = field(default_factory=ImageEncoderCfg)
    text_decoder: TextDecoderCfg = field(default_factory=TextDecoderCfg)


_SUBTYPES = {('ModelCfg', 'image_encoder'): ImageEncoderCfg, ('ModelCfg', 'text_decoder'): TextDecoderCfg}


def _scan_model_configs():
    global _MODEL_CONFIGS
    files = []
    for p in _MODEL_CONFIG_PATHS:
        if p.is_file() and p.suffix == '.json':
            files.append(p)
        elif p.is_dir():
            files.extend(p.glob('*.json'))
    for cf in files:
        _MODEL_CONFIGS[cf.stem] = ModelCfg.load(cf)
    _MODEL_CONFIGS = {k: v for k, v in sorted(_MODEL_CONFIGS.items(), key=lambda x: _natural_key(x[0]))}


_scan_model_configs()


def list_models():
    return list(_MODEL_CONFIGS.keys())


def get_model_config(model_name):
    model_name = clean_name(model_name)
    return copy.deepcopy(_MODEL_CONFIGS.get(model_name, None))
