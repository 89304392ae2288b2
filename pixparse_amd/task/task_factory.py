"""name -> (TaskCls, CfgCls) registry (ref: task/task_factory.py:44-79).  The pretrain task is the hot path; the
fine-tune tasks reuse its step, the eval tasks (f-4) the KV-cache generation path; `donut_eval_ocr` (a
different model family) is not built and raises a clear error."""
from ..framework import DeviceEnv, Monitor
from .task_cruller_pretrain import TaskCrullerPretrain, TaskCrullerPretrainCfg
from .task_cruller_eval_ocr import TaskCrullerEvalOCR, TaskCrullerEvalOCRCfg
from .task_cruller_eval_rvlcdip import TaskCrullerEvalRVLCDIP, TaskCrullerEvalRVLCDIPCfg
from .task_cruller_eval_docvqa import TaskCrullerEvalCORD, TaskCrullerEvalCORDCfg, TaskCrullerEvalDOCVQA, TaskCrullerEvalDOCVQACfg
from .task_cruller_finetune import (TaskCrullerFinetuneCORD, TaskCrullerFinetuneCORDCfg, TaskCrullerFinetuneDOCVQA,
                                    TaskCrullerFinetuneDOCVQACfg, TaskCrullerFinetuneRVLCDIP, TaskCrullerFinetuneRVLCDIPCfg)

from .task_cruller_finetune_xent import TaskCrullerFinetuneXent, TaskCrullerFinetuneXentCfg

_NOT_BUILT = ('donut_eval_ocr',)


class TaskFactory:
    TASK_CLASS_REGISTRY = {
        'cruller_pretrain': (TaskCrullerPretrain, TaskCrullerPretrainCfg),
        'cruller_eval_ocr': (TaskCrullerEvalOCR, TaskCrullerEvalOCRCfg),
        'cruller_eval_rvlcdip': (TaskCrullerEvalRVLCDIP, TaskCrullerEvalRVLCDIPCfg),
        'cruller_eval_cord': (TaskCrullerEvalCORD, TaskCrullerEvalCORDCfg),
        'cruller_eval_docvqa': (TaskCrullerEvalDOCVQA, TaskCrullerEvalDOCVQACfg),
        'cruller_finetune_rvlcdip': (TaskCrullerFinetuneRVLCDIP, TaskCrullerFinetuneRVLCDIPCfg),
        'cruller_finetune_cord': (TaskCrullerFinetuneCORD, TaskCrullerFinetuneCORDCfg),
        'cruller_finetune_docvqa': (TaskCrullerFinetuneDOCVQA, TaskCrullerFinetuneDOCVQACfg),
        'cruller_finetune_xent': (TaskCrullerFinetuneXent, TaskCrullerFinetuneXentCfg),
    }

    @classmethod
    def create_task(cls, task_name: str, task_args, device_env: DeviceEnv, monitor: Monitor):
        task_name = task_name.lower()
        if task_name in _NOT_BUILT:
            raise NotImplementedError(f'task {task_name!r} exists in the reference but is outside the MI355X hot-path scope '
                                      '(SURVEY.md §8f); built: ' + ', '.join(cls.TASK_CLASS_REGISTRY))
        if task_name not in cls.TASK_CLASS_REGISTRY:
            raise ValueError(f'Unknown task type: {task_name}. Available tasks are {list(cls.TASK_CLASS_REGISTRY.keys())}')
        task_cls, task_cfg = cls.TASK_CLASS_REGISTRY[task_name]
        task_cfg_instance = task_cfg(**(vars(task_args) if not isinstance(task_args, dict) else task_args))
        task_cls_instance = task_cls(cfg=task_cfg_instance, device_env=device_env, monitor=monitor)
        return task_cls_instance, task_cfg_instance
