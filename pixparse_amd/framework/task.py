"""Task base classes (ref: framework/task.py:9-90): counters + hook names of the plug-in surface."""
from typing import Any, Dict

from .config import TaskEvalCfg, TaskTrainCfg
from .device import DeviceEnv
from .monitor import Monitor


class Task:
    def __init__(self, device_env: DeviceEnv, monitor: Monitor = None):
        self.device_env = device_env
        self.monitor = monitor


class TaskEval(Task):
    def __init__(self, cfg: TaskEvalCfg, device_env: DeviceEnv, monitor: Monitor = None):
        super().__init__(device_env=device_env, monitor=monitor)

    def collate_fn(self, batch):
        pass

    def setup(self, *args, **kwargs):
        pass

    def prepare_for_evaluation(self):
        pass

    def step(self, sample: Dict[str, Any]) -> Dict[str, Any]:
        pass

    def end(self):
        pass


class TaskTrain(Task):
    def __init__(self, cfg: TaskTrainCfg, device_env: DeviceEnv, monitor: Monitor = None):
        super().__init__(device_env=device_env, monitor=monitor)
        self.num_intervals = cfg.num_intervals
        self.num_warmup_intervals = cfg.num_warmup_intervals
        self.eval_frequency = cfg.eval_frequency
        self.num_steps_per_interval = None
        self.start_interval = 0
        self.step = 0                 # optimizer update count
        self.batch_idx = 0            # total train batch count
        self.interval_idx = 0
        self.interval_batch_idx = 0
        self.optimizer = None
        self.scheduler = None
        self.scaler = None
        self.autocast = None

    def collate_fn(self, batch):
        pass

    def train_setup(self, *args, **kwargs):
        pass

    def train_interval_start(self):
        pass

    def train_interval_end(self):
        pass

    def train_step(self, sample: Dict[str, Any]) -> Dict[str, Any]:
        pass

    def eval_step(self, sample: Dict[str, Any]) -> Dict[str, Any]:
        pass

    def get_current_lr(self):
        lrl = [param_group['lr'] for param_group in self.optimizer.param_groups]
        return sum(lrl) / len(lrl)
