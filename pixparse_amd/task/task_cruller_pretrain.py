"""Cruller pretraining task -- the plug-in on the hot path (ref: task/task_cruller_pretrain.py).

Same constructor / ``train_setup`` / ``train_interval_start|end`` / ``train_step(sample)`` /
``state_dict`` surface and the same counters as the reference; adds the ``forward()`` method the
reference only has as a closure (task_cruller_pretrain.py:247-257).  Underneath, one micro-step is:
H2D -> token shift -> Cruller.forward_loss (HIP engines + fused CE) -> Cruller.backward with bucketed
RCCL all-reduce overlapped -> on update steps the two-kernel optimiser tail (unscale/clip-norm/AdamW/
zero_grad/bf16-shadow refresh) -> cosine LR update.  No host synchronisation inside a step.
"""
import logging
from dataclasses import dataclass, field
from functools import partial
from typing import Optional

import torch

from ..data import preprocess_ocr_anno, preprocess_text_anno
from ..framework import DeviceEnv, Monitor, TaskTrain, TaskTrainCfg
from ..framework.optim import ArenaAdamW, CosineLRScheduler, LossScaler
from ..framework.reducer import BucketedGradReducer
from ..models import Cruller, ModelCfg, get_model_config
from ..tokenizers import TokenizerCfg, TokenizerHF

_logger = logging.getLogger(__name__)


@dataclass
class TaskCrullerPretrainCfg(TaskTrainCfg):
    model_name: Optional[str] = None
    model: ModelCfg = field(default_factory=ModelCfg)
    tokenizer: TokenizerCfg = field(default_factory=TokenizerCfg)
    # not a reference field: the reference's decoder drops hidden states (p = 0.1) exactly when it was built with pretrained=False
    # (from_config leaves train mode); here that behaviour is an explicit switch, off by default
    decoder_dropout: bool = False
    # not a reference field: replay the whole micro-step (forward, CE, backward, optimiser tail) from a hipGraph captured on the second
    # step of each kind.  None = automatic: on for single-process runs of launch-bound models (< 250 M parameters: cfg-1 / cfg-2 spend
    # their step in hundreds of microsecond-scale launches), off when the step has host-visible state (dropout masks, RCCL buckets) or
    # PIXPARSE_AMD_GRAPH_STEP=0 says so; PIXPARSE_AMD_GRAPH_STEP=1 forces it on where it is legal
    graph_step: Optional[bool] = None

    def __post_init__(self):
        if self.model_name:
            model = get_model_config(self.model_name)
            if model is None:
                _logger.warning(f'Model config for {self.model_name} was not found, using defaults.')
            else:
                self.model = model
        else:
            self.model_name = 'custom'


class ImagePreprocess:
    """ToTensor -> bicubic antialiased Resize(image_size) -> Normalize (ref :132-143), in plain torch (CPU, loader side)."""

    def __init__(self, image_size, mean, std, num_chs, grayscale=False):
        self.image_size, self.num_chs, self.grayscale = tuple(image_size), num_chs, grayscale
        self.mean = torch.tensor(mean if isinstance(mean, (tuple, list)) else [mean], dtype=torch.float32).view(-1, 1, 1)
        self.std = torch.tensor(std if isinstance(std, (tuple, list)) else [std], dtype=torch.float32).view(-1, 1, 1)

    def __call__(self, img):
        import numpy as np
        if not isinstance(img, torch.Tensor):
            a = np.asarray(img)
            if a.ndim == 2:
                a = a[:, :, None]
            img = torch.from_numpy(a).permute(2, 0, 1).float() / 255.0
        if self.grayscale and img.shape[0] == 3:   # torchvision Grayscale(): ITU-R 601-2 luma
            img = (0.2989 * img[0] + 0.587 * img[1] + 0.114 * img[2])[None]
        x = torch.nn.functional.interpolate(img[None], size=self.image_size, mode='bicubic', antialias=True, align_corners=False)[0]
        return (x - self.mean) / self.std


class TaskCrullerPretrain(TaskTrain):
    def __init__(self, cfg: TaskCrullerPretrainCfg, device_env: DeviceEnv, monitor: Monitor = None):
        super().__init__(cfg=cfg, device_env=device_env, monitor=monitor)
        self.cfg = cfg
        if cfg.dtype not in (None, 'bfloat16', 'bf16'):
            raise NotImplementedError(f'dtype {cfg.dtype!r}: only the bfloat16 autocast policy is implemented on MI355X')
        if not cfg.amp:
            raise NotImplementedError('amp=False (pure fp32) is not implemented: the HIP path is the bf16 autocast policy')
        self.amp_dtype = torch.bfloat16

        self.task_start_token = '<s_pretrain>'
        self.prompt_end_token = self.task_start_token
        self.max_position_embeddings = cfg.model.text_decoder.max_length
        self.text_anno_fn = False
        self.tokenizer = TokenizerHF(cfg.tokenizer)
        special_tokens = ['<sep/>', self.task_start_token, self.prompt_end_token]
        newly_added_num = self.tokenizer.trunk.add_special_tokens({'additional_special_tokens': sorted(set(special_tokens))})
        self.vocab_size = len(self.tokenizer.trunk)

        preproc_fn = preprocess_text_anno if self.text_anno_fn else preprocess_ocr_anno
        self.anno_preprocess_train = partial(preproc_fn, tokenizer=self.tokenizer.trunk,
                                             max_position_embeddings=self.max_position_embeddings,
                                             task_start_token=self.task_start_token, prompt_end_token=self.prompt_end_token)

        self.model = Cruller(cfg.model)
        if newly_added_num > 0:
            self.model.text_decoder.trunk.resize_token_embeddings(len(self.tokenizer.trunk))

        self.has_no_sync = False
        self.num_image_chs = 1 if cfg.model.image_encoder.image_fmt == 'L' else 3
        img_mean = self.model.image_encoder.trunk.pretrained_cfg['mean']
        img_std = self.model.image_encoder.trunk.pretrained_cfg['std']
        self.img_mean = sum(img_mean) / len(img_mean) if cfg.model.image_encoder.image_fmt == 'L' else img_mean
        self.img_std = sum(img_std) / len(img_std) if cfg.model.image_encoder.image_fmt == 'L' else img_std
        self.image_preprocess_train = ImagePreprocess(cfg.model.image_encoder.image_size, self.img_mean, self.img_std, self.num_image_chs)
        self.image_preprocess_eval = None
        self.train_metrics = {}
        self.eval_metrics = {}
        self.max_recursion_length = 1000
        self.reducer = None
        self.last_loss = None

    # ------------------------------------------------------------------ setup
    def train_setup(self, num_batches_per_interval: int):
        device = self.device_env.device
        if device.type != 'cuda':
            raise RuntimeError('TaskCrullerPretrain.train_setup: an MI355X (cuda/HIP device) is required; no CPU path exists')
        self.model.to(device)
        self.model.set_train_dropout(bool(self.cfg.decoder_dropout), seed=42 + self.device_env.global_rank)   # random_seed(42, rank) convention
        opt = self.cfg.opt
        if opt.optimizer != 'adamw':
            raise NotImplementedError(f'optimizer {opt.optimizer!r}: only adamw (the reference default) is implemented')
        if opt.scheduler != 'cosine':
            raise NotImplementedError(f'scheduler {opt.scheduler!r}: only cosine is implemented')
        if opt.layer_decay is not None or opt.momentum is not None:
            raise NotImplementedError('layer_decay / momentum are not supported by the fused AdamW path')
        self.clip_mode = (opt.clip_grad_mode or 'norm') if opt.clip_grad_value is not None else None    # timm dispatch_clip_grad (ref :270-278)
        if self.clip_mode not in (None, 'norm', 'value'):
            raise NotImplementedError(f'clip_grad_mode {opt.clip_grad_mode!r}: "norm" and "value" are implemented (not "agc")')
        kw = {}
        if opt.betas is not None:
            kw['betas'] = tuple(opt.betas)
        # NOTE weight_decay is not forwarded, exactly like the reference (:196-203) -> 0
        self.optimizer = ArenaAdamW(self.model.arena, lr=opt.learning_rate, eps=opt.eps, weight_decay=0.0, **kw)
        self.optimizer.set_clip_value(opt.clip_grad_value if self.clip_mode == 'value' else None)
        self.model._ensure_engines()
        self.reducer = BucketedGradReducer(self.model.arena, self.device_env.world_size,
                                           active=getattr(self.device_env, 'distributed', self.device_env.world_size > 1))
        if self.reducer.active:
            self.reducer.broadcast_params(0)
            self.model.refresh_shadows(full=True)
            self.has_no_sync = True
        self.scaler = LossScaler(enabled=True).attach(self.optimizer.state)
        self.autocast = None
        self.num_steps_per_interval = num_batches_per_interval // opt.grad_accum_steps
        self.scheduler = CosineLRScheduler(self.optimizer, t_initial=self.num_intervals * self.num_steps_per_interval,
                                           warmup_t=self.num_warmup_intervals * self.num_steps_per_interval,
                                           warmup_lr_init=opt.warmup_learning_rate)
        self.scheduler.step_update(0)
        # the persistent GEMMs' wave-quantisation model: built-in constants (reproducible bits); PIXPARSE_AMD_GEMM_CALIBRATE=1 refits it to what THIS device sustains
        from .. import ops as _ops
        self.gemm_model = _ops.gemm_calibrate(self.device_env.device)
        # hipGraph replay of the micro-step: legal when nothing the step launches depends on host state -- the LR and the bias
        # corrections come from device words (crl_optim_prepare), the GradScaler lives on the device, batches are copied into static buffers
        import os
        want = self.cfg.graph_step
        env = os.environ.get('PIXPARSE_AMD_GRAPH_STEP')
        if want is None:
            want = (env == '1') or (env != '0' and self.model.arena.total < 250_000_000)
        legal = not self.reducer.active and getattr(self.model, '_drop', None) is None
        if want and not legal and (self.cfg.graph_step or env == '1'):
            _logger.warning('graph_step requested but the step has host-visible state (data-parallel buckets or dropout masks): running eagerly')
        self._graph_on = bool(want and legal)
        self._graphs = {}          # need_update -> 'warm' | torch.cuda.CUDAGraph
        self._graph_in = None      # static input buffers (image, shifted tokens, shifted targets)

    def train_interval_start(self):
        self.optimizer.zero_grad()
        self.interval_batch_idx = 0

    def train_interval_end(self):
        if self.monitor is not None:
            self.monitor.log_phase('train', self.interval_idx)
        self.interval_idx += 1

    # ------------------------------------------------------------------ the step
    def forward(self, image_input, text_input, text_target):
        """autocast(bf16){ model -> logits -> CE(ignore_index=-100) } / grad_accum_steps  (ref :247-257).
        inputs already on device and shifted; returns the device scalar loss and leaves dloss/dlogits ready."""
        accum = self.cfg.opt.grad_accum_steps
        return self.model.forward_loss(image_input, text_input, text_target, loss_mul=1.0 / accum, grad_mul=1.0 / accum,
                                       grad_mul_dev=self.scaler.scale_tensor())

    def _backward(self, need_update: bool, first_micro: bool = False):
        self.reducer.enabled = need_update or not self.has_no_sync
        self.reducer.begin()
        self.model.backward(self.reducer.on_ready if self.reducer.active else None, first_micro=first_micro)
        self.reducer.finish()
        if need_update:
            opt = self.cfg.opt
            # NativeScaler.__call__ (ref :259-278): unscale_ -> clip_grad_norm_ -> scaler.step (skips on inf/nan) -> scaler.update
            self.optimizer.step(clip_norm=opt.clip_grad_value if self.clip_mode == 'norm' else None, zero_grad=True, scaler=self.scaler,
                                grad_divisor=self.reducer.grad_divisor())
            self.model.refresh_shadows(full=False)

    def _graphed_micro_step(self, image_input, text_input, text_target, need_update: bool, first_micro: bool = False):
        """forward + CE + backward (+ optimiser tail on update steps) replayed from a hipGraph.  The batch is copied into static device
        buffers; the first micro-step of each kind (with / without the optimiser tail) runs eagerly (it sizes every activation buffer
        and scratch), the second is captured and replayed, all later ones are replays.  Same kernels, same arguments, same order as the
        eager step: losses and parameters are bit-identical (tests/test_model_gpu.py::test_graphed_train_step_equals_eager)."""
        shapes = (tuple(image_input.shape), tuple(text_input.shape), tuple(text_target.shape))
        if self._graph_in is None:
            dev = self.device_env.device
            self._graph_in = (torch.empty(shapes[0], dtype=torch.float32, device=dev), torch.empty(shapes[1], dtype=text_input.dtype, device=dev),
                              torch.empty(shapes[2], dtype=text_target.dtype, device=dev))
        gi, gt, gy = self._graph_in
        if shapes != (tuple(gi.shape), tuple(gt.shape), tuple(gy.shape)):      # a ragged last batch: run it eagerly
            loss = self.forward(image_input, text_input, text_target)
            self._backward(need_update, first_micro)
            return loss
        gi.copy_(image_input, non_blocking=True)
        gt.copy_(text_input, non_blocking=True)
        gy.copy_(text_target, non_blocking=True)
        kind = (need_update, first_micro)       # the captured kernels differ: optimiser tail, overwrite / accumulate weight gradients
        st = self._graphs.get(kind)
        if st is None:
            loss = self.forward(gi, gt, gy)
            self._backward(need_update, first_micro)
            self._graphs[kind] = 'warm'
            return loss
        if st == 'warm':
            torch.cuda.synchronize()
            from .. import ops
            ops.freeze_scratch()       # the capture bakes scratch addresses into the graph: they must stay owned by the scratch objects
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                self.forward(gi, gt, gy)
                self._backward(need_update, first_micro)
            self._graphs[kind] = st = graph
        st.replay()
        return self.model._loss

    log_phase_name = 'train'

    def train_step(self, sample):
        image_input, text_input, text_target = sample
        return self._train_step_shifted(image_input, text_input[:, :-1], text_target[:, 1:])

    def _train_step_shifted(self, image_input, text_input, text_target):
        """one micro-step on decoder inputs / labels that are already shifted against each other"""
        result = {}
        device = self.device_env.device
        image_input = image_input.to(device, non_blocking=True)
        text_input = text_input.to(device, non_blocking=True)
        text_target = text_target.to(device, non_blocking=True)

        accum_steps = self.cfg.opt.grad_accum_steps
        need_update = (self.interval_batch_idx + 1) % accum_steps == 0
        # the first micro-step behind a zero-filled gradient arena (train_interval_start / the AdamW kernel's fused zero fill)
        first_micro = self.interval_batch_idx % accum_steps == 0
        if getattr(self, '_graph_on', False):
            loss = self._graphed_micro_step(image_input, text_input, text_target, need_update, first_micro)
        else:
            loss = self.forward(image_input, text_input, text_target)
            self._backward(need_update, first_micro)
        self.last_loss = loss

        self.batch_idx += 1
        self.interval_batch_idx += 1
        if not need_update:
            return result
        self.step += 1
        self.scheduler.step_update(self.step)
        # optimizer.zero_grad() is fused into the AdamW kernel (zero_grad=True above)
        if self.step % self.eval_frequency == 0 and self.monitor is not None:
            # the reference's train-time OCR metric call is broken (missing prompt_token arg): loss/lr only
            self.monitor.log_step(self.log_phase_name, step_idx=self.step, step_end_idx=self.num_intervals * self.num_steps_per_interval,
                                  interval=self.interval_idx, loss=loss.item(), lr=self.get_current_lr(), metrics=self.train_metrics)
        return result

    def state_dict(self):
        sd = {'model': self.model.state_dict(), 'optimizer': self.optimizer.state_dict()}
        if hasattr(self.scheduler, 'state_dict'):
            sd['scheduler'] = self.scheduler.state_dict()
        if self.scaler is not None:
            sd['scaler'] = self.scaler.state_dict()
        return sd

    def load_state_dict(self, state_dict):
        pass  # like the reference (:382-383)

    # ---- full resume (the reference saves model weights only and cannot resume pretraining, Q5)
    def training_state(self):
        """state_dict() + the counters needed to continue the run exactly"""
        sd = self.state_dict()
        sd['counters'] = dict(step=self.step, batch_idx=self.batch_idx, interval_idx=self.interval_idx,
                              interval_batch_idx=self.interval_batch_idx)
        return sd

    def load_training_state(self, sd):
        """call after train_setup(); accepts DDP-style 'module.' prefixed model keys (ref app/eval.py:135)"""
        model_sd = {(k[7:] if k.startswith('module.') else k): v for k, v in sd['model'].items()}
        self.model.load_state_dict(model_sd)
        self.model.refresh_shadows(full=True)
        self.optimizer.load_state_dict(sd['optimizer'])
        if 'scaler' in sd and self.scaler is not None:
            self.scaler.load_state_dict(sd['scaler'])
        for k, v in sd.get('counters', {}).items():
            setattr(self, k, v)
        self.scheduler.step_update(self.step)
        self.optimizer.set_update_count(self.step)        # the device-side clock of the LR schedule

    def __repr__(self):
        return '\n'.join([f'model: {self.model.cfg}', f'opt: {repr(self.optimizer)}', f'sched: {repr(self.scheduler)}'])
