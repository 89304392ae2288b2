"""Classification fine-tune of the image encoder (ref: task/task_cruller_finetune_xent.py, registry name `cruller_finetune_xent`).

The reference loads a pretraining checkpoint into Cruller, then trains ``nn.Sequential(encoder, GetCLSToken, nn.Linear(768, 16))`` on
RVL-CDIP labels with CrossEntropyLoss under the same NativeScaler / clip / AdamW / cosine machinery as the pretraining task (:139-318).
Here the head joins the model's parameter arena (models/cruller.py: add_classifier_head) and one micro-step is encoder forward ->
token 0 -> head GEMM -> fused CE -> head backward -> encoder backward -> the two-kernel optimiser tail: the HIP engines of the
pretraining step, minus the decoder.
"""
from collections import OrderedDict
from dataclasses import dataclass
from typing import Any, Dict

import torch

from ..framework import DeviceEnv, Monitor
from .task_cruller_pretrain import TaskCrullerPretrain, TaskCrullerPretrainCfg

NUM_CLASSES = 16      # RVL-CDIP (ref :147)
HEAD_FEATURES = 768   # the reference hard-codes nn.Linear(768, 16): a ViT-base sized encoder


@dataclass
class TaskCrullerFinetuneXentCfg(TaskCrullerPretrainCfg):
    pass


class _Classifier:
    """what `task.model` is after train_setup() in the reference: an nn.Sequential with the keys `encoder.*` / `final_fc.*`"""

    def __init__(self, cruller):
        self.cruller = cruller

    def __call__(self, image_input):
        return self.cruller.classify(image_input)[:, :NUM_CLASSES]

    def _names(self):
        for k in self.cruller.arena.entries:
            if k.startswith('image_encoder.'):
                yield k, 'encoder.' + k[len('image_encoder.'):]
            elif k.startswith('final_fc.'):
                yield k, k

    def state_dict(self):
        sd = self.cruller.state_dict()
        return OrderedDict((new, sd[old]) for old, new in self._names())

    def load_state_dict(self, sd, strict=True):
        full = self.cruller.state_dict()
        back = {new: old for old, new in self._names()}
        missing = [k for k in back if k not in sd]
        unexpected = [k for k in sd if k not in back]
        if strict and (missing or unexpected):
            raise RuntimeError(f'classifier state dict: missing {missing[:3]}, unexpected {unexpected[:3]}')
        for k, v in sd.items():
            if k in back:
                full[back[k]] = v
        self.cruller.load_state_dict(full)

    def parameters(self):
        return [self.cruller._pmap[old] for old, _ in self._names()]

    def named_parameters(self):
        return [(new, self.cruller._pmap[old]) for old, new in self._names()]

    def __getattr__(self, name):      # arena, refresh_shadows, device, ...: the machinery of the pretraining task keeps working
        return getattr(self.cruller, name)


class TaskCrullerFinetuneXent(TaskCrullerPretrain):
    log_phase_name = 'finetune'

    def __init__(self, cfg: TaskCrullerFinetuneXentCfg, device_env: DeviceEnv, monitor: Monitor = None):
        super().__init__(cfg, device_env, monitor)
        # ref :69-70, :82-89: the task token is <s_finetune>; the tokenizer / annotation preprocessing are built but unused by the step
        old = self.task_start_token
        self.task_start_token = self.prompt_end_token = '<s_finetune>'
        added = self.tokenizer.trunk.add_special_tokens({'additional_special_tokens': sorted({'<sep/>', self.task_start_token})})
        if added > 0:
            self.model.text_decoder.trunk.resize_token_embeddings(len(self.tokenizer.trunk))
        self.vocab_size = len(self.tokenizer.trunk)
        if old != self.task_start_token:
            self.anno_preprocess_train.keywords.update(task_start_token=self.task_start_token, prompt_end_token=self.prompt_end_token)
        # app/train.py assigns the checkpoint to `task.state_dict` and sets `task.resume` (ref app/train.py:156-157): the instance attribute
        # shadows the method until train_setup() has consumed it (same convention as task_cruller_finetune.py)
        self.resume = False

    def train_setup(self, num_batches_per_interval: int):
        ckpt = self.__dict__.get('state_dict')
        if self.resume and ckpt is not None:                   # ref :141-144
            self.model.load_state_dict({k.replace('module.', ''): v for k, v in ckpt.items()})
        self.__dict__.pop('state_dict', None)                  # give the method back (checkpoints need it)
        self.model.add_classifier_head(NUM_CLASSES, HEAD_FEATURES)      # ref :146-151
        super().train_setup(num_batches_per_interval)
        self._graph_on = False                                 # the graphed micro-step of the base class takes (image, text, target)
        self.classifier = _Classifier(self.model)

    def collate_fn(self, batch):
        """PIL images + integer labels, as the RVL-CDIP loader returns them (ref :206-217)"""
        images = torch.stack([self.image_preprocess_train(item['image']) for item in batch])
        labels = torch.tensor([item['label'] for item in batch], dtype=torch.int64)
        return {'image': images, 'label': labels}

    def forward(self, image_input, label, _unused=None):
        accum = self.cfg.opt.grad_accum_steps
        return self.model.classify_loss(image_input, label, loss_mul=1.0 / accum, grad_mul=1.0 / accum, grad_mul_dev=self.scaler.scale_tensor())

    def _backward(self, need_update: bool):
        self.reducer.enabled = need_update or not self.has_no_sync
        self.reducer.begin()
        self.model.classify_backward(self.reducer.on_ready if self.reducer.active else None)
        self.reducer.finish()
        if need_update:
            opt = self.cfg.opt
            self.optimizer.step(clip_norm=opt.clip_grad_value if self.clip_mode == 'norm' else None, zero_grad=True, scaler=self.scaler,
                                grad_divisor=self.reducer.grad_divisor())
            self.model.refresh_shadows(full=False)

    def train_step(self, sample: Dict[str, Any]) -> Dict[str, Any]:
        device = self.device_env.device
        image_input = sample['image'].to(device, non_blocking=True)
        label = sample['label'].to(device, non_blocking=True)
        accum_steps = self.cfg.opt.grad_accum_steps
        need_update = (self.interval_batch_idx + 1) % accum_steps == 0
        loss = self.forward(image_input, label)
        self._backward(need_update)
        self.last_loss = loss
        self.batch_idx += 1
        self.interval_batch_idx += 1
        if not need_update:
            return {}
        self.step += 1
        self.scheduler.step_update(self.step)
        if self.step % self.eval_frequency == 0 and self.monitor is not None:
            self.monitor.log_step(self.log_phase_name, step_idx=self.step, step_end_idx=self.num_intervals * self.num_steps_per_interval,
                                  interval=self.interval_idx, loss=loss.item(), lr=self.get_current_lr(), metrics=None, eval_data=None)
        return {}

    def eval_step(self, sample):
        pass   # like the reference (:310-312)

    def state_dict(self):
        sd = super().state_dict()
        if getattr(self, 'classifier', None) is not None:
            sd['model'] = self.classifier.state_dict()       # the keys the reference's Sequential saves: encoder.* / final_fc.*
        return sd
