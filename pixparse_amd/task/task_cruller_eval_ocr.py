"""Cruller OCR evaluation task (ref: task/task_cruller_eval_ocr.py).
`setup()` loads the checkpoint handed over as `task.resume_state_dict` (ref app/eval.py:124-139), `step(sample)` encodes
the page images, generates greedily through the KV-cache decode path (utils/ocr_utils.py) and scores CER / WER against
the targets, `average_metrics` folds the per-batch numbers exactly like the reference (:229-240)."""
import logging
import time
from dataclasses import dataclass, field
from functools import partial
from typing import Optional

import torch

from ..data import preprocess_text_anno
from ..framework import DeviceEnv, Monitor, TaskEval, TaskEvalCfg
from ..models import Cruller, ModelCfg, get_model_config
from ..tokenizers import TokenizerCfg, TokenizerHF
from ..utils.ocr_utils import get_ocr_metrics
from .task_cruller_pretrain import ImagePreprocess

_logger = logging.getLogger(__name__)


@dataclass
class TaskCrullerEvalOCRCfg(TaskEvalCfg):
    model_name: Optional[str] = None
    model: ModelCfg = field(default_factory=ModelCfg)
    tokenizer: TokenizerCfg = field(default_factory=TokenizerCfg)

    def __post_init__(self):
        if self.model_name:
            model = get_model_config(self.model_name)
            if model is None:
                _logger.warning(f'Model config for {self.model_name} was not found, using defaults.')
            else:
                self.model = model
        else:
            self.model_name = 'custom'


def time_and_log(func):
    def wrapper(self, *args, **kwargs):
        t0 = time.time()
        result = func(self, *args, **kwargs)
        _logger.info(f'Executed method {func.__name__} in {time.time() - t0:.2f} seconds')
        return result
    return wrapper


class TaskCrullerEvalOCR(TaskEval):
    def __init__(self, cfg: TaskCrullerEvalOCRCfg, device_env: DeviceEnv, monitor: Monitor = None):
        super().__init__(cfg=cfg, device_env=device_env, monitor=monitor)
        self.cfg = cfg
        self.amp_dtype = torch.bfloat16
        self.task_start_token = '<s_pretrain>'
        self.prompt_end_token = self.task_start_token
        self.max_position_embeddings = cfg.model.text_decoder.max_length
        self.text_anno_fn = True
        self.tokenizer = TokenizerHF(cfg.tokenizer)
        special_tokens = ['<sep/>', self.task_start_token, self.prompt_end_token]
        newly_added_num = self.tokenizer.trunk.add_special_tokens({'additional_special_tokens': sorted(set(special_tokens))})
        self.vocab_size = len(self.tokenizer.trunk)
        self.anno_preprocess_eval = partial(preprocess_text_anno, tokenizer=self.tokenizer.trunk,
                                            max_position_embeddings=self.max_position_embeddings,
                                            task_start_token=self.task_start_token, prompt_end_token=self.prompt_end_token)
        self.model = Cruller(cfg.model)
        if newly_added_num > 0:
            self.model.text_decoder.trunk.resize_token_embeddings(len(self.tokenizer.trunk))
        self.has_no_sync = False
        self.num_image_chs = 1 if cfg.model.image_encoder.image_fmt == 'L' else 3
        img_mean = self.model.image_encoder.trunk.pretrained_cfg['mean']
        img_std = self.model.image_encoder.trunk.pretrained_cfg['std']
        self.img_mean = sum(img_mean) / len(img_mean) if self.num_image_chs == 1 else img_mean
        self.img_std = sum(img_std) / len(img_std) if self.num_image_chs == 1 else img_std
        self.image_preprocess_eval = ImagePreprocess(cfg.model.image_encoder.image_size, self.img_mean, self.img_std, self.num_image_chs,
                                                     grayscale=self.num_image_chs == 1)   # RGB pages on an 'L' model are folded to luma
        self.eval_metrics = {}
        self.max_recursion_length = 1000
        self.resume_state_dict = None

    def setup(self):
        """weights arrive through `task.resume_state_dict` (model state dict, optional DDP 'module.' prefixes)"""
        device = self.device_env.device
        if device.type != 'cuda':
            raise RuntimeError('TaskCrullerEvalOCR.setup: an MI355X (cuda/HIP device) is required; no CPU path exists')
        if self.resume_state_dict:
            self.model.load_state_dict({k.replace('module.', ''): v for k, v in self.resume_state_dict.items()})
        self.model.eval()
        self.model.to(device)
        self.model._ensure_engines()
        self.model.refresh_shadows(full=True)

    def prepare_for_evaluation(self, loaders: dict) -> dict:
        return {k: v for k, v in loaders.items() if k in ['eval', 'eval_FUNSD']}

    @time_and_log
    def step(self, sample):
        metrics = {}
        image_input, text_input, text_target = sample
        if isinstance(text_target, (list, tuple)):           # loader hands lists of per-page tensors (ref :201-207)
            text_target = torch.stack([item[0] if isinstance(item, (list, tuple)) else item for item in text_target], dim=0)
        device = self.device_env.device
        ocr_metrics, _ = get_ocr_metrics(model=self.model, tokenizer=self.tokenizer, image_input=image_input.to(device, non_blocking=True),
                                         text_input=text_target.to(device, non_blocking=True), device_env=self.device_env,
                                         max_recursion_length=self.max_recursion_length, prompt_token=self.task_start_token)
        metrics['ocr_reconstruction'] = ocr_metrics
        return metrics

    def average_metrics(self, metrics: dict):
        wer_sum = sum(m['ocr_reconstruction']['wer'] for m in metrics.values())
        cer_sum = sum(m['ocr_reconstruction']['cer'] for m in metrics.values())
        n = len(metrics)
        return {'ocr_reconstruction': {'wer': wer_sum / n, 'cer': cer_sum / n}}

    def end(self):
        pass

    def state_dict(self):
        return {'model': self.model.state_dict()}
