"""Training entry point with the reference's shape (ref: app/train.py:25-192): TrainCfg / task cfg / data cfg parsed from
`--train.*`, `--task.*`, `--task.opt.*`, `--data.train.*` flags (dash or underscore spelling, like simple_parsing's
DASH variants), DeviceEnv, TaskFactory.create_task, per-interval loop, `checkpoint-{i}.pt` = model.state_dict().

Data: `--data.train.source` (tar shards / directories of page images + .json annotations, ref app/train.py:171-181 ->
data/loader.py) goes through data.create_loader with the task's image / annotation preprocessing; `--data.train.gpu-preprocess
true` moves the resize + normalise onto the GPU.  Without a source a synthetic loader of the same shapes is used:

    python -m pixparse_amd.app.train --task.model-name cruller_large_1280x960 --task.dtype bfloat16 \
        --task.opt.learning-rate 3e-4 --task.opt.clip-grad-value 1.0 --task.opt.clip-grad-mode norm \
        --data.train.batch-size 8 --data.train.num-batches 20 --train.num-intervals 2 --train.output-dir /tmp/out
    python -m pixparse_amd.app.train ... --data.train.source '/data/shards/docs-{000..127}.tar' --data.train.num-samples 100000 \
        --data.train.batch-size 8 --data.train.num-workers 8 --data.train.gpu-preprocess true
"""
import argparse
import dataclasses
import logging
import os
import sys
import time
from dataclasses import dataclass, field, fields, is_dataclass
from datetime import datetime
from typing import Optional, get_type_hints

import torch

from ..data import SyntheticLoaderBundle
from ..framework import DeviceEnv, Monitor, OptimizationCfg, random_seed, setup_logging, train_one_interval
from ..task import TaskCrullerPretrainCfg, TaskFactory

_logger = logging.getLogger('train')


@dataclass
class TrainCfg:
    """ref: app/train.py:25-44"""
    experiment: Optional[str] = None
    output_dir: str = './output'
    log_filename: str = 'out.log'
    resume: bool = False
    checkpoint_path: str = ''
    output_checkpoint_dir: Optional[str] = None
    seed: int = 42
    task_name: str = 'cruller_pretrain'
    num_intervals: Optional[int] = None   # overrides task.num_intervals when given
    save_checkpoints: bool = True


@dataclass
class SyntheticDataCfg:
    """--data.train.*: DatasetCfg fields (ref data/config.py:12-20) when `source` is given, else the synthetic loader's"""
    batch_size: int = 8
    num_batches: int = 10     # synthetic: batches per interval
    ragged: bool = False
    source: Optional[str] = None
    num_samples: Optional[int] = None
    split: str = 'train'
    format: str = 'webdataset'
    num_workers: int = 4
    gpu_preprocess: bool = False


def _add_flags(parser, prefix, cls):
    hints = get_type_hints(cls)
    for f in fields(cls):
        t = hints[f.name]
        if is_dataclass(t):
            _add_flags(parser, prefix + f.name + '.', t)
            continue
        names = {f'--{prefix}{f.name}', f'--{prefix}{f.name}'.replace('_', '-')}
        parser.add_argument(*sorted(names), dest=prefix + f.name, default=None)


def _coerce(value, t):
    origin = getattr(t, '__origin__', None)
    if origin is not None:
        args = [a for a in t.__args__ if a is not type(None)]
        if value in ('None', 'none'):
            return None
        if origin is tuple or (args and getattr(args[0], '__origin__', None) is tuple):
            inner = args[0].__args__ if origin is not tuple else t.__args__
            parts = [p for p in value.replace(',', ' ').split() if p]
            return tuple(inner[0](p) for p in parts)
        return _coerce(value, args[0])
    if t is bool:
        return value.lower() in ('1', 'true', 'yes')
    return t(value)


def _build(cls, prefix, ns):
    hints = get_type_hints(cls)
    kw = {}
    for f in fields(cls):
        t = hints[f.name]
        if is_dataclass(t):
            kw[f.name] = _build(t, prefix + f.name + '.', ns)
            continue
        v = getattr(ns, prefix + f.name)
        if v is not None:
            kw[f.name] = _coerce(v, t)
    return cls(**kw)


def parse_args(argv=None):
    parser = argparse.ArgumentParser(description='pixparse_amd training (MI355X)')
    _add_flags(parser, 'train.', TrainCfg)
    _add_flags(parser, 'task.', TaskCrullerPretrainCfg)
    _add_flags(parser, 'data.train.', SyntheticDataCfg)
    ns = parser.parse_args(argv)
    return _build(TrainCfg, 'train.', ns), _build(TaskCrullerPretrainCfg, 'task.', ns), _build(SyntheticDataCfg, 'data.train.', ns)


def train(cfg: TrainCfg, task, loader, checkpoint_dir):
    """ref: app/train.py:47-67"""
    device_env = task.device_env
    for i in range(task.start_interval, task.num_intervals):
        loader.set_interval(i)
        t0 = time.time()
        train_one_interval(task, loader)
        if device_env.device.type == 'cuda':
            torch.cuda.synchronize()
        dt = time.time() - t0
        if device_env.is_primary():
            _logger.info(f'interval {i}: {loader.num_samples * device_env.world_size / dt:.2f} docs/s, loss {float(task.last_loss):.5f}')
            if cfg.save_checkpoints:
                torch.save(task.model.state_dict(), os.path.join(checkpoint_dir, f'checkpoint-{i}.pt'))


def main(argv=None):
    train_cfg, task_cfg, data_cfg = parse_args(argv)
    if train_cfg.num_intervals is not None:
        task_cfg.num_intervals = train_cfg.num_intervals
    device_env = DeviceEnv()
    random_seed(train_cfg.seed, 0)   # before model construction so the init is reproducible
    task, task_cfg = TaskFactory.create_task(train_cfg.task_name, task_cfg.__dict__, device_env, None)
    random_seed(train_cfg.seed, rank=device_env.global_rank)
    if train_cfg.experiment is None:
        date_str = device_env.broadcast_object(datetime.now().strftime('%Y%m%d-%H%M%S'))
        train_cfg.experiment = '-'.join([date_str, f'task_{train_cfg.task_name}', f'model_{task_cfg.model_name}'])
    experiment_path = os.path.join(train_cfg.output_dir, train_cfg.experiment)
    checkpoint_dir = train_cfg.output_checkpoint_dir or os.path.join(experiment_path, 'checkpoints')
    log_path = None
    if device_env.is_primary():
        os.makedirs(experiment_path, exist_ok=True)
        os.makedirs(checkpoint_dir, exist_ok=True)
        log_path = os.path.join(experiment_path, train_cfg.log_filename)
    setup_logging(log_path)
    task.monitor = Monitor(train_cfg.experiment, output_dir=experiment_path, output_enabled=device_env.is_primary())
    if train_cfg.resume:
        sd = torch.load(train_cfg.checkpoint_path, map_location='cpu')
        sd = sd.get('model', sd)
        task.model.load_state_dict({k[7:] if k.startswith('module.') else k: v for k, v in sd.items()})   # ref app/eval.py:135
    m = task.model
    if data_cfg.source:
        # ref app/train.py:171-181: the task supplies the preprocessing, the loader the IO
        from ..data import DatasetCfg, DeviceImagePreprocess, create_loader
        assert data_cfg.num_samples, '--data.train.num-samples (samples per interval, all ranks) is required with a source'
        ds_cfg = DatasetCfg(source=data_cfg.source, num_samples=data_cfg.num_samples, batch_size=data_cfg.batch_size, split=data_cfg.split,
                            format=data_cfg.format, num_workers=data_cfg.num_workers, gpu_preprocess=data_cfg.gpu_preprocess)
        image_preprocess = task.image_preprocess_train
        if ds_cfg.gpu_preprocess:
            image_preprocess = DeviceImagePreprocess(m.img_size, task.img_mean, task.img_std, task.num_image_chs, device_env.device)
        loader = create_loader(ds_cfg, is_train=True, collate_fn=getattr(task, 'collate_fn', None), image_preprocess=image_preprocess,
                               anno_preprocess=task.anno_preprocess_train, image_fmt=task_cfg.model.image_encoder.image_fmt,
                               seed=train_cfg.seed, world_size=device_env.world_size, global_rank=device_env.global_rank)
    else:
        loader = SyntheticLoaderBundle(batch_size=data_cfg.batch_size, num_batches=data_cfg.num_batches, in_chans=m.in_chans,
                                       img_size=m.img_size, max_length=m.max_length, vocab_size=task.vocab_size, seed=train_cfg.seed,
                                       rank=device_env.global_rank, ragged=data_cfg.ragged)
    task.train_setup(num_batches_per_interval=loader.num_batches)
    if device_env.is_primary():
        _logger.info(task)
    train(train_cfg, task, loader, checkpoint_dir)


if __name__ == '__main__':
    main()
