from .preprocess import preprocess_ocr_anno, preprocess_text_anno, mask_targets
from .synthetic import SyntheticLoaderBundle, synthetic_batch
from .gpu_preprocess import GpuImagePreprocess, aa_bicubic_tables
from .config import DataCfg, DatasetCfg
from .loader import DeviceImagePreprocess, LoaderBundle, create_doc_anno_pipe, create_loader
