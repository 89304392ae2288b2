"""RVL-CDIP classification eval (ref: task/task_cruller_eval_rvlcdip.py): the class is read off the
first tokens generated after `<s_rvlcdip>`. The reference re-runs the whole decoder on the re-tokenised strings for each
of its 5 steps (:266-311); here the same 5 greedy steps go through the KV-cache decode path. Counting rule kept as is:
a sample scores when, at ANY generated `</s>`, the text accumulated so far (task / bos / eos tokens removed, stripped)
equals `<label/>` -- at most once per sample."""
import logging
from dataclasses import dataclass

import torch

from ..framework import DeviceEnv, Monitor
from .task_cruller_eval_ocr import TaskCrullerEvalOCR, TaskCrullerEvalOCRCfg
from .task_cruller_finetune import TaskCrullerFinetuneRVLCDIP

_logger = logging.getLogger(__name__)


@dataclass
class TaskCrullerEvalRVLCDIPCfg(TaskCrullerEvalOCRCfg):
    pass


class TaskCrullerEvalRVLCDIP(TaskCrullerEvalOCR):
    MAX_STEPS = 5   # "Few steps for RVL CDIP, we have to predict at most 3 tokens" (ref :264)

    def __init__(self, cfg: TaskCrullerEvalRVLCDIPCfg, device_env: DeviceEnv, monitor: Monitor = None):
        super().__init__(cfg, device_env, monitor)      # tokenizer + model with the two pretrain tokens
        self.task_start_token = '<s_rvlcdip>'
        self.prompt_end_token = self.task_start_token
        special_tokens = ['<sep/>', self.task_start_token, self.prompt_end_token, *TaskCrullerFinetuneRVLCDIP.DATASET_TOKENS]
        if self.tokenizer.trunk.add_special_tokens({'additional_special_tokens': sorted(set(special_tokens))}) > 0:
            self.model.text_decoder.trunk.resize_token_embeddings(len(self.tokenizer.trunk))
        self.vocab_size = len(self.tokenizer.trunk)
        self.int2str = dict(enumerate(TaskCrullerFinetuneRVLCDIP.CLASSES))

    def collate_fn(self, batch):
        """PIL / uint8 pages + integer labels -> {'image', 'label'}; unreadable images are dropped (ref :218-241)"""
        items = [it for it in batch if it is not None]
        if not items:
            return None
        images, labels = [], []
        for it in items:
            try:
                images.append(self.image_preprocess_eval(it['image']))
                labels.append(int(it['label']))
            except Exception as e:   # the reference filters PIL.UnidentifiedImageError
                _logger.info(f'Encountered image issue {e}. Filtering...')
        return {'image': torch.stack(images), 'label': torch.tensor(labels, dtype=torch.int64)}

    def step(self, sample):
        ground_truths = [self.int2str[int(gt)] for gt in sample['label']]
        n = len(ground_truths)
        counted = [False] * n
        strings = ['<s_rvlcdip>'] * n
        correct = 0
        tok = self.tokenizer.trunk
        with torch.inference_mode():
            images = torch.stack([im for im in sample['image']]).to(self.device_env.device)
            enc = self.model.image_encoder(images)
            self.model.decode_begin(enc, self.MAX_STEPS + 1)
            ids = torch.full((n, 1), tok.convert_tokens_to_ids('<s_rvlcdip>'), dtype=torch.int64, device=self.device_env.device)
            for _ in range(self.MAX_STEPS):
                ids = torch.argmax(self.model.decode_step(ids), dim=-1, keepdim=True)
                for i, t in enumerate(ids[:, 0].tolist()):
                    piece = tok.decode([t])
                    strings[i] += piece
                    if piece == '</s>':
                        label = strings[i].replace('<s_rvlcdip>', '').replace('</s>', '').replace('<s>', '').strip()
                        if label == '<' + ground_truths[i] + '/>' and not counted[i]:
                            correct += 1
                            counted[i] = True
        return {'classification': {'correct_samples': correct, 'n_valid_samples': n}}

    def average_metrics(self, metrics: dict):
        correct = sum(m['classification']['correct_samples'] for m in metrics.values())
        total = sum(m['classification']['n_valid_samples'] for m in metrics.values())
        return {'classification': {'accuracy': correct / total}}
