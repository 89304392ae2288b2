"""GPU half of the loader boundary: pages travel as uint8 and the HIP resize + normalise kernel builds
the batch (DeviceImagePreprocess) -- equal to the CPU Compose path of the same loader; `python -m pixparse_amd.app.train
--data.train.source ...` end to end over a tiny on-disk shard set (ref app/train.py:171-188)."""
import os

import pytest
import torch

from .test_loader_cpu import _make_docs, _task_fns, _write_tar

pytestmark = pytest.mark.gpu


def test_device_preprocess_loader_matches_cpu_loader(tmp_path):
    from pixparse_amd.data import DatasetCfg, DeviceImagePreprocess, create_loader
    dev = torch.device('cuda:0')
    docs = _make_docs(16, seed=4)
    _write_tar(str(tmp_path / 'd.tar'), docs)
    img_pre, anno_pre, _ = _task_fns(size=(48, 40))
    cfg = DatasetCfg(source=str(tmp_path / 'd.tar'), num_samples=16, batch_size=4, num_workers=2)
    cpu = create_loader(cfg, is_train=True, image_preprocess=img_pre, anno_preprocess=anno_pre, image_fmt='L', seed=9)
    gpu = create_loader(cfg, is_train=True, image_preprocess=DeviceImagePreprocess((48, 40), 0.5, 0.25, 1, dev), anno_preprocess=anno_pre,
                        image_fmt='L', seed=9)
    n = 0
    for (ic, tc, gc), (ig, tg, gg) in zip(cpu.loader, gpu.loader):
        assert ig.is_cuda and ig.shape == (4, 1, 48, 40) and ig.dtype == torch.float32
        assert torch.equal(tc, tg) and torch.equal(gc, gg)
        assert float((ig.cpu() - ic).abs().max()) < 2e-5
        n += 1
    assert n == 4


def test_app_train_over_on_disk_shards(tmp_path):
    """cruller_small (swin_tiny 224x224 RGB + BART-base 2L), 2 intervals x 3 batches from tar shards, GPU preprocessing,
    checkpoints written; the loss moves"""
    from pixparse_amd.app.train import main
    docs = _make_docs(12, seed=5)
    for s in range(2):
        _write_tar(str(tmp_path / f'docs-{s:03d}.tar'), docs[s::2])
    out = str(tmp_path / 'out')
    main(['--task.model-name', 'cruller_small', '--task.dtype', 'bfloat16', '--task.opt.learning-rate', '1e-3', '--task.opt.clip-grad-value', '1.0',
          '--task.opt.clip-grad-mode', 'norm', '--task.num-warmup-intervals', '0', '--task.tokenizer.name', 'byte-bart',
          '--data.train.source', str(tmp_path / 'docs-{000..001}.tar'), '--data.train.num-samples', '6', '--data.train.batch-size', '2',
          '--data.train.num-workers', '2', '--data.train.gpu-preprocess', 'true',
          '--train.num-intervals', '2', '--train.output-dir', out, '--train.experiment', 'exp'])
    ck = os.path.join(out, 'exp', 'checkpoints')
    assert sorted(os.listdir(ck)) == ['checkpoint-0.pt', 'checkpoint-1.pt']
    sd = torch.load(os.path.join(ck, 'checkpoint-1.pt'), map_location='cpu')
    assert all(torch.isfinite(v).all() for v in sd.values())
