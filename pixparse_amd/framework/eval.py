"""evaluation driver (ref: framework/eval.py:4-25): every loader the task accepts is swept batch by batch through
`task.step`; if the task can average its per-batch metrics, only the average is kept under `metrics[key]["average"]`."""
from .task import TaskEval


def evaluate(task: TaskEval, loaders: dict) -> dict:
    metrics = {}
    for key, loader in task.prepare_for_evaluation(loaders).items():
        per_batch = {i: task.step(sample) for i, sample in enumerate(loader.loader)}
        metrics[key] = {'average': task.average_metrics(per_batch)} if hasattr(task, 'average_metrics') else per_batch
    return metrics
