"""Dataset config dataclasses (ref: data/config.py:12-25); `gpu_preprocess` is the one MI355X-specific addition."""
from dataclasses import dataclass
from typing import Optional


@dataclass
class DatasetCfg:
    source: str
    num_samples: int
    batch_size: int
    split: str = 'train'            # "train", "test", "val" (hf_dataset only)
    format: str = 'webdataset'      # "webdataset" (tar shards / directories of page + json) or "hf_dataset"
    num_workers: int = 4
    gpu_preprocess: bool = False    # workers only decode; resize + normalise run on the GPU (crl_image_preprocess_u8)


@dataclass
class DataCfg:
    train: Optional[DatasetCfg] = None
    eval: Optional[DatasetCfg] = None
