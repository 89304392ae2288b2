"""Optimiser tail of the train step over the flat arenas (ref: task/task_cruller_pretrain.py:191-224,
259-295 -> timm create_optimizer_v2('adamw') = torch.optim.AdamW(weight_decay=0), timm NativeScaler =
torch GradScaler, dispatch_clip_grad('norm') = clip_grad_norm_, timm CosineLRScheduler).

Everything device-side runs in two launches over contiguous memory: crl_grad_norm_scaled (sum of squares ->
norm, inf check, clip coefficient, GradScaler.update(), step counter: all left in device memory) and crl_adamw
(unscale * clip, AdamW with bias corrections from the device-side step count, zero_grad and the bf16 shadow refresh
fused).  No host synchronisation anywhere in a step.
"""
import math
from typing import Optional

import torch

from .. import ops


STATE_FLOATS = 16  # [0] grad norm, [1] clip * unscale coefficient, [2] found inf/nan, [3] optimiser steps TAKEN,
                   # [4] GradScaler loss scale, [5] GradScaler growth tracker, [6] clip-by-value threshold (0 = off), [7] updates ATTEMPTED
                   # (the scheduler's clock), [8] learning rate of this update, [9] 1 - beta1^t, [10] 1 / sqrt(1 - beta2^t), [11] / [12] the
                   # exact INTEGER counts behind [3] / [7] (uint32 bit patterns: fp32 counters stop at 2^24)   (include/crl.h)


class ArenaAdamW:
    """torch.optim.AdamW semantics (single param group, decoupled weight decay) on a ParamArena.  The step counter that
    feeds the bias corrections lives on the device (state[3]) and advances only when the step is taken: a step that
    GradScaler skips (inf / nan gradients) leaves it unchanged, exactly like torch's `state['step']`."""

    def __init__(self, arena, lr: float, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0):
        self.arena = arena
        self.param_groups = [dict(lr=lr, initial_lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay)]
        arena.alloc_training_state()
        self.state = torch.zeros(STATE_FLOATS, dtype=torch.float32, device=arena.p.device)
        self.state[4] = 1.0   # no loss scaling until a LossScaler attaches
        self.schedule = None  # (warmup_lr_init, lr_min, warmup_t, t_initial) of an attached CosineLRScheduler: the LR is then computed on the device

    def set_clip_value(self, value: Optional[float]):
        """timm dispatch_clip_grad mode 'value' (torch clip_grad_value_): clamp every unscaled gradient element to [-value, value]
        inside the AdamW launch; None / 0 switches it off"""
        self.state[6] = float(value or 0.0)

    @property
    def step_count(self) -> int:
        """optimiser steps actually taken (reads the device counter: a host sync, for checkpoints / tests only)"""
        return int(self.state[11:12].view(torch.int32).item())

    def zero_grad(self, set_to_none: bool = False):
        self.arena.g.zero_()

    def step(self, clip_norm: Optional[float] = None, inv_scale: Optional[float] = None, zero_grad: bool = False,
             scaler: Optional['LossScaler'] = None, grad_divisor: float = 1.0):
        """unscale / inf check / clip-norm coefficient, then AdamW (skipped on inf / nan).  With `scaler` the loss scale is
        the device word state[4] and GradScaler.update() happens inside the same launch (crl_grad_norm_scaled);
        without it `inv_scale` is a host number (plain crl_grad_norm)."""
        g = self.param_groups[0]
        a = self.arena
        mx = clip_norm if clip_norm is not None else 0.0
        if scaler is not None and scaler.enabled:
            ops.grad_norm_scaled(a.g, mx, grad_divisor, scaler.growth_factor, scaler.backoff_factor, scaler.growth_interval, self.state)
        else:
            ops.grad_norm(a.g, mx, (1.0 if inv_scale is None else inv_scale) / (grad_divisor if inv_scale is None else 1.0), self.state)
        # learning rate + bias corrections on the device (crl_optim_prepare): no launch argument of the step changes from step to step.
        # With a scheduler attached the LR follows its closed form from the device-side update counter (g['lr'] is the host's mirror of it,
        # for logging); without one g['lr'] is passed as the constant rate.
        if self.schedule is not None:
            w0, lr_min, warmup_t, t_initial = self.schedule
            ops.optim_prepare(self.state, g['initial_lr'], w0, lr_min, warmup_t, t_initial, g['betas'][0], g['betas'][1])
        else:
            ops.optim_prepare(self.state, g['lr'], 0.0, 0.0, 0, 0, g['betas'][0], g['betas'][1])
        ops.adamw(a.p, a.g, a.m, a.v, a.pb, -1.0, g['betas'][0], g['betas'][1], g['eps'], g['weight_decay'], 0, self.state, zero_grad)

    def grad_norm(self) -> torch.Tensor:
        """device scalar: unscaled global L2 norm seen by the last step()"""
        return self.state[0]

    def found_inf(self) -> torch.Tensor:
        """device scalar: 1.0 when the last step() saw inf / nan gradients and was skipped"""
        return self.state[2]

    def state_dict(self):
        return dict(step=self.step_count, param_groups=[dict(g) for g in self.param_groups],
                    exp_avg=self.arena.m.clone(), exp_avg_sq=self.arena.v.clone())

    def set_update_count(self, n: int):
        """the scheduler's clock (updates attempted so far), e.g. after a resume"""
        self.state[7] = float(n)
        self.state[12:13].view(torch.int32).fill_(int(n))

    def load_state_dict(self, sd):
        self.state[3] = float(sd['step'])
        self.state[11:12].view(torch.int32).fill_(int(sd['step']))
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
        if hasattr(optimizer, 'schedule'):      # ArenaAdamW evaluates the same closed form on the device (include/crl.h crl_optim_prepare)
            optimizer.schedule = (float(warmup_lr_init), float(lr_min), int(warmup_t), int(t_initial))

    def _get_lr(self, t: int):
        if t < self.warmup_t:
            return [self.warmup_lr_init + t * (v - self.warmup_lr_init) / self.warmup_t for v in self.base_values]
        if t < self.t_initial:
            return [self.lr_min + 0.5 * (v - self.lr_min) * (1 + math.cos(math.pi * t / self.t_initial)) for v in self.base_values]
        return [self.lr_min for _ in self.base_values]

    def step_update(self, num_updates: int, metric=None):
        """host mirror of the rate the next update will use (logging, get_current_lr); the device clock advances by itself with every
        update -- after a resume call optimizer.set_update_count(num_updates) as well"""
        for g, lr in zip(self.optimizer.param_groups, self._get_lr(num_updates)):
            g['lr'] = lr

    def state_dict(self):
        return {k: v for k, v in self.__dict__.items() if k != 'optimizer'}

    def __repr__(self):
        return f'CosineLRScheduler(t_initial={self.t_initial}, warmup_t={self.warmup_t}, warmup_lr_init={self.warmup_lr_init})'


class LossScaler:
    """torch.amp.GradScaler (init 65536, x2 every 2000 clean steps, x0.5 on inf / nan) with the scale and the growth
    tracker resident on the DEVICE (words 4 and 5 of the optimiser's state vector): the cross-entropy kernel multiplies
    the gradient by the scale it reads there, crl_grad_norm_scaled unscales with the same word, decides found-inf and
    applies GradScaler.update() -- the step after an overflow already runs with the halved scale, with no host
    synchronisation anywhere.  With bf16 the scale is a power of two and cancels exactly."""

    def __init__(self, init_scale=65536.0, growth_factor=2.0, backoff_factor=0.5, growth_interval=2000, enabled=True):
        self._init_scale = float(init_scale) if enabled else 1.0
        self.growth_factor, self.backoff_factor, self.growth_interval = float(growth_factor), float(backoff_factor), int(growth_interval)
        self.enabled = enabled
        self._state = None         # the optimiser's device state vector once attached
        self._host = [self._init_scale, 0.0]

    def attach(self, state: torch.Tensor):
        """bind to the optimiser's state vector (>= 8 floats); the current scale / tracker move there"""
        assert state.numel() >= 16
        state[4] = self._host[0]
        state[5] = self._host[1]
        self._state = state
        return self

    def scale_tensor(self) -> Optional[torch.Tensor]:
        """device fp32 scalar holding the loss scale (None while detached or disabled)"""
        return self._state[4:5] if (self._state is not None and self.enabled) else None

    def get_scale(self) -> float:
        """host copy of the scale (synchronises when attached: logging / checkpoints only)"""
        return float(self._state[4]) if self._state is not None else self._host[0]

    def get_growth_tracker(self) -> int:
        return int(round(float(self._state[5]))) if self._state is not None else int(self._host[1])

    def state_dict(self):
        return dict(scale=self.get_scale(), growth_factor=self.growth_factor, backoff_factor=self.backoff_factor,
                    growth_interval=self.growth_interval, _growth_tracker=self.get_growth_tracker())

    def load_state_dict(self, sd):
        self._host = [float(sd['scale']), float(sd.get('_growth_tracker', 0))]
        if self._state is not None:
            self._state[4] = self._host[0]
            self._state[5] = self._host[1]
