from .config import OptimizationCfg, TaskEvalCfg, TaskTrainCfg
from .device import DeviceEnv, DeviceEnvType, is_distributed_env, world_info_from_env
from .logger import setup_logging
from .monitor import Monitor
from .random import random_seed
from .task import Task, TaskEval, TaskTrain
from .train import train_one_interval
from .eval import evaluate
