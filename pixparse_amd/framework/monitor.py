"""Minimal Monitor with the reference's call shape (framework/monitor.py:164-256): text lines +
optional CSV summary.  TensorBoard / W&B are out of scope (observability only)."""
import csv
import logging
import os
import time
from typing import Optional

_logger = logging.getLogger(__name__)


class Monitor:
    def __init__(self, experiment_name=None, output_dir=None, logger=None, hparams=None, wandb=False,
                 tensorboard=False, output_enabled=True, log_eval_data=False):
        self.output_dir = output_dir
        self.logger = logger or _logger
        self.output_enabled = output_enabled
        self.csv_path = os.path.join(output_dir, 'summary.csv') if output_dir else None
        self.step_idx = 0
        self._t_last = None
        self.rows = []

    def log_step(self, phase: str, step_idx: int, step_end_idx: Optional[int] = None, interval: Optional[int] = None,
                 loss: Optional[float] = None, rate=None, lr=None, phase_suffix: str = '', metrics: dict = None,
                 eval_data: dict = None, **kwargs):
        if not self.output_enabled:
            return
        now = time.time()
        dt = None if self._t_last is None else now - self._t_last
        self._t_last = now
        msg = f'{phase.title()}{phase_suffix}: step {step_idx}' + (f'/{step_end_idx}' if step_end_idx else '')
        if interval is not None:
            msg += f' interval {interval}'
        if loss is not None:
            msg += f' loss: {loss:.6f}'
        if lr is not None:
            msg += f' lr: {lr:.3e}'
        if rate is not None:
            msg += f' rate: {rate:.2f}/s'
        if dt is not None:
            msg += f' ({dt:.2f}s since last log)'
        self.logger.info(msg)
        self.rows.append(dict(phase=phase, step=step_idx, interval=interval, loss=loss, lr=lr))

    def log_phase(self, phase: str = 'eval', interval: Optional[int] = None, name_map: dict = None, **kwargs):
        if not self.output_enabled:
            return
        self.logger.info(f'{phase.title()} interval {interval} done. ' + ' '.join(f'{k}: {v}' for k, v in kwargs.items()))
        if self.csv_path and self.rows:
            new = not os.path.exists(self.csv_path)
            with open(self.csv_path, 'a') as f:
                w = csv.DictWriter(f, fieldnames=list(self.rows[0].keys()))
                if new:
                    w.writeheader()
                w.writerows(self.rows)
            self.rows = []
