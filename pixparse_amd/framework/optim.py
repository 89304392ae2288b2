"""Optimiser tail of the train step over the flat arenas (ref: task/task_cruller_pretrain.py:191-224,
259-295 -> timm create_optimizer_v2('adamw') = torch.optim.AdamW(weight_decay=0), timm NativeScaler =
torch GradScaler, dispatch_clip_grad('norm') = clip_grad_norm_, timm CosineLRScheduler).

Everything device-side runs in two launches over contiguous memory: crl_grad_norm (sum of squares ->
norm, inf check, clip coefficient, all left in device memory) and crl_adamw (unscale * clip, AdamW,
zero_grad and the bf16 shadow refresh fused).  No host synchronisation anywhere in a step.
"""
import math
from typing import Optional

import torch

from .. import ops


class ArenaAdamW:
    """torch.optim.AdamW semantics (single param group, decoupled weight decay) on a ParamArena."""

    def __init__(self, arena, lr: float, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0):
        self.arena = arena
        self.param_groups = [dict(lr=lr, initial_lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay)]
        self.step_count = 0
        arena.alloc_training_state()
        self.state = torch.zeros(4, dtype=torch.float32, device=arena.p.device)  # [norm, coef, found_inf, -]

    def zero_grad(self, set_to_none: bool = False):
        self.arena.g.zero_()

    def step(self, clip_norm: Optional[float] = None, inv_scale: float = 1.0, zero_grad: bool = False):
        g = self.param_groups[0]
        a = self.arena
        ops.grad_norm(a.g, clip_norm if clip_norm is not None else 0.0, inv_scale, self.state)
        self.step_count += 1
        ops.adamw(a.p, a.g, a.m, a.v, a.pb, g['lr'], g['betas'][0], g['betas'][1], g['eps'], g['weight_decay'],
                  self.step_count, self.state, zero_grad)

    def grad_norm(self) -> torch.Tensor:
        """device scalar: unscaled global L2 norm seen by the last step()"""
        return self.state[0]

    def state_dict(self):
        return dict(step=self.step_count, param_groups=[dict(g) for g in self.param_groups],
                    exp_avg=self.arena.m.clone(), exp_avg_sq=self.arena.v.clone())

    def load_state_dict(self, sd):
        self.step_count = sd['step']
        self.param_groups = [dict(g) for g in sd['param_groups']]
        self.arena.m.copy_(sd['exp_avg'])
        self.arena.v.copy_(sd['exp_avg_sq'])

    def __repr__(self):
        g = self.param_groups[0]
        return f"ArenaAdamW(lr={g['lr']}, betas={g['betas']}, eps={g['eps']}, weight_decay={g['weight_decay']}, n={self.arena.total})"


class CosineLRScheduler:
    """timm CosineLRScheduler(t_initial, lr_min=0, warmup_t, warmup_lr_init, t_in_epochs=False, cycle_limit=1)."""

    def __init__(self, optimizer, t_initial: int, warmup_t: int = 0, warmup_lr_init: float = 0.0, lr_min: float = 0.0):
        self.optimizer = optimizer
        self.t_initial, self.warmup_t, self.warmup_lr_init, self.lr_min = t_initial, warmup_t, warmup_lr_init, lr_min
        self.base_values = [g['initial_lr'] for g in optimizer.param_groups]

    def _get_lr(self, t: int):
        if t < self.warmup_t:
            return [self.warmup_lr_init + t * (v - self.warmup_lr_init) / self.warmup_t for v in self.base_values]
        if t < self.t_initial:
            return [self.lr_min + 0.5 * (v - self.lr_min) * (1 + math.cos(math.pi * t / self.t_initial)) for v in self.base_values]
        return [self.lr_min for _ in self.base_values]

    def step_update(self, num_updates: int, metric=None):
        for g, lr in zip(self.optimizer.param_groups, self._get_lr(num_updates)):
            g['lr'] = lr

    def state_dict(self):
        return {k: v for k, v in self.__dict__.items() if k != 'optimizer'}

    def __repr__(self):
        return f'CosineLRScheduler(t_initial={self.t_initial}, warmup_t={self.warmup_t}, warmup_lr_init={self.warmup_lr_init})'


class LossScaler:
    """torch.amp.GradScaler bookkeeping (init 65536, x2 every 2000 clean steps, x0.5 on inf/nan) without host
    syncs: the found-inf flag of step t is copied to pinned memory asynchronously and folded into the scale
    when it has arrived (normally before step t+1).  With bf16 the scale is a power of two and cancels exactly."""

    def __init__(self, init_scale=65536.0, growth_factor=2.0, backoff_factor=0.5, growth_interval=2000, enabled=True):
        self.scale = float(init_scale) if enabled else 1.0
        self.growth_factor, self.backoff_factor, self.growth_interval = growth_factor, backoff_factor, growth_interval
        self.enabled = enabled
        self._growth_tracker = 0
        self._pending = []  # (event, pinned tensor)

    def get_scale(self):
        return self.scale

    def note_step(self, state: torch.Tensor):
        if not self.enabled or not state.is_cuda:
            return
        host = torch.empty(1, dtype=torch.float32, pin_memory=True)
        host.copy_(state[2:3], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._pending.append((ev, host))

    def update(self):
        still = []
        for ev, host in self._pending:
            if ev.query():
                if float(host[0]) != 0.0:
                    self.scale *= self.backoff_factor
                    self._growth_tracker = 0
                else:
                    self._growth_tracker += 1
                    if self._growth_tracker == self.growth_interval:
                        self.scale *= self.growth_factor
                        self._growth_tracker = 0
            else:
                still.append((ev, host))
        self._pending = still

    def state_dict(self):
        return dict(scale=self.scale, growth_factor=self.growth_factor, backoff_factor=self.backoff_factor,
                    growth_interval=self.growth_interval, _growth_tracker=self._growth_tracker)

    def load_state_dict(self, sd):
        self.scale = sd['scale']
        self._growth_tracker = sd.get('_growth_tracker', 0)
