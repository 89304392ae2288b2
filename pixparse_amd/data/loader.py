"""Real-data loader boundary of the train step (ref: data/loader.py:24-119, data/config.py:12-25,
app/train.py:171-183).

`create_loader(cfg, is_train, image_preprocess, anno_preprocess, collate_fn, image_key, image_fmt, start_interval, seed,
world_size, global_rank, create_decoder_pipe)` keeps the reference signature and returns a
`LoaderBundle(loader, num_batches, num_samples, sampler)` with `.set_interval(i)` -- everything `app/train.py` and
`framework/train.py` touch.  The reference delegates the work to `chug` (`create_wds_loader`, `create_doc_anno_pipe`:
an un-vendored, un-pinned dependency that is not installed here -- parity with chug itself is UNPINNED); what is
restated is the contract the task relies on:

  * a "webdataset" source is a set of tar shards (or plain directories) whose members are grouped by basename into
    samples: one page image (`image_key` lists the accepted extensions in priority order) + one `.json` annotation;
  * the decoder pipe turns a sample into  image = image_preprocess(PIL page in `image_fmt`)  and
    (text, target) = anno_preprocess(annotation)  -- the task's own callables (task_cruller_pretrain.py:104-110,132-143;
    `preprocess_ocr_anno` picks the page, so the same page of a multi-frame TIFF is decoded);
  * training batches are `(image [B, C, H, W], text_input [B, L], text_target [B, L])` -- what `train_step` unpacks
    (task_cruller_pretrain.py:237-242); evaluation batches keep the per-page lists (task_cruller_eval_ocr.py:199-205);
  * samples that fail to decode are skipped with a warning, ranks / workers read disjoint parts, training reshuffles per
    interval (`set_interval`), `num_batches = num_samples // (batch_size * world_size)`.

MI355X-specific: when `image_preprocess.on_device` is set (data.gpu_preprocess.GpuImagePreprocess / DeviceImagePreprocess)
the workers only DECODE -- pages travel as uint8 HWC tensors (1/4 of the fp32 bytes, no CPU resize) and the main process
runs the HIP resize + normalise kernel into the device batch: the CPU Compose of the reference leaves the critical path.
"""
import io
import json
import logging
import os
import random
import tarfile
from dataclasses import dataclass
from functools import partial
from glob import glob
from typing import Callable, Dict, Iterator, List, Optional, Tuple

import torch
from torch.utils.data import DataLoader, DistributedSampler, IterableDataset, get_worker_info

from .config import DatasetCfg

_logger = logging.getLogger(__name__)
DEFAULT_IMAGE_KEY = 'pdf;tif;tiff;png;jpg;jpeg'


@dataclass
class LoaderBundle:
    """chug.common.LoaderBundle: what app/train.py and framework/train.py use of a loader"""
    loader: object
    num_batches: int
    num_samples: int
    sampler: Optional[object] = None

    def set_interval(self, interval: int):
        if self.sampler is not None and hasattr(self.sampler, 'set_epoch'):
            self.sampler.set_epoch(interval)
        if hasattr(self.loader, 'set_interval'):
            self.loader.set_interval(interval)


# ------------------------------------------------------------------------------------------------ shards -> raw samples
def expand_source(source: str) -> List[str]:
    """'dir', 'a.tar', 'shard-{000..003}.tar', 'shards/*.tar', or several of them joined by '::' -> list of shard paths"""
    out = []
    for part in source.split('::'):
        part = part.strip()
        if '{' in part and '..' in part:
            pre, rest = part.split('{', 1)
            rng, post = rest.split('}', 1)
            lo, hi = rng.split('..')
            out += [f'{pre}{i:0{len(lo)}d}{post}' for i in range(int(lo), int(hi) + 1)]
        elif any(ch in part for ch in '*?['):
            out += sorted(glob(part))
        else:
            out.append(part)
    if not out:
        raise FileNotFoundError(f'no shards match {source!r}')
    return out


def _split_key(name: str) -> Tuple[str, str]:
    base = os.path.basename(name)
    stem, _, ext = base.partition('.')
    return os.path.join(os.path.dirname(name), stem), ext.lower()


def iter_shard(path: str) -> Iterator[Dict[str, bytes]]:
    """raw samples {'__key__': ..., ext: bytes} of one shard (tar file or directory), members grouped by basename"""
    if os.path.isdir(path):
        groups: Dict[str, Dict[str, str]] = {}
        for root, _, files in sorted(os.walk(path)):
            for fn in sorted(files):
                key, ext = _split_key(os.path.join(root, fn))
                groups.setdefault(key, {})[ext] = os.path.join(root, fn)
        for key in sorted(groups):
            sample = {'__key__': key}
            for ext, fp in groups[key].items():
                with open(fp, 'rb') as f:
                    sample[ext] = f.read()
            yield sample
        return
    with tarfile.open(path, 'r:*') as tf:      # webdataset convention: the members of a sample are adjacent
        cur_key, sample = None, None
        for m in tf:
            if not m.isfile():
                continue
            key, ext = _split_key(m.name)
            if key != cur_key:
                if sample is not None:
                    yield sample
                cur_key, sample = key, {'__key__': key}
            sample[ext] = tf.extractfile(m).read()
        if sample is not None:
            yield sample


# ------------------------------------------------------------------------------------------------ decoder pipe
def create_doc_anno_pipe(image_preprocess: Callable, anno_preprocess: Callable, image_key: str = DEFAULT_IMAGE_KEY,
                         image_fmt: str = 'L', seed: int = 0) -> Callable:
    """raw sample -> dict(image, text, target, meta); raises on samples that cannot be used (the loader skips them)"""
    exts = [e.strip().lower() for e in image_key.split(';') if e.strip()]
    on_device = bool(getattr(image_preprocess, 'on_device', False))

    def decode(sample: Dict[str, bytes], rng: random.Random):
        from PIL import Image
        if 'json' not in sample:
            raise KeyError('no .json annotation')
        anno = json.loads(sample['json'])
        res = anno_preprocess(anno, generator=rng)
        tok, meta = res if isinstance(res, tuple) else (res, {})
        ext = next((e for e in exts if e in sample), None)
        if ext is None:
            raise KeyError(f'no image member ({image_key})')
        if ext == 'pdf':
            raise NotImplementedError('pdf pages need a rasteriser that is not available in this image')
        img = Image.open(io.BytesIO(sample[ext]))
        page = (meta.get('page_indices') or [0])[0]
        if getattr(img, 'n_frames', 1) > 1:
            img.seek(min(page, img.n_frames - 1))
        img = img.convert(image_fmt)
        if on_device:
            import numpy as np
            a = np.asarray(img)
            image = torch.from_numpy(a[:, :, None].copy() if a.ndim == 2 else a.copy())      # uint8 [H, W, C]
        else:
            image = image_preprocess(img)
        return dict(image=image, text=tok['text'], target=tok['target'], meta=meta, key=sample.get('__key__'))
    return decode


def collate_doc_pages(samples: List[dict], is_train: bool):
    """train: (image [B, C, H, W] | list of uint8 pages, text [B, L], target [B, L]) -- one page per document
    (preprocess_ocr_anno yields min(1, num_pages) pages);  eval: per-document lists of per-page tensors"""
    images = [s['image'] for s in samples]
    image = torch.stack(images) if images[0].dtype != torch.uint8 else images
    if is_train:
        return image, torch.stack([s['text'][0] for s in samples]), torch.stack([s['target'][0] for s in samples])
    return image, [s['text'] for s in samples], [s['target'] for s in samples]


class _DocShards(IterableDataset):
    """yields collated batches; shards (or, with fewer shards than readers, samples) are split over ranks x workers"""

    def __init__(self, shards, decoder, batch_size, num_batches, is_train, seed, world_size, global_rank, collate_fn):
        self.shards, self.decoder = shards, decoder
        self.batch_size, self.num_batches, self.is_train = batch_size, num_batches, is_train
        self.seed, self.world_size, self.global_rank = seed, world_size, global_rank
        self.collate_fn = collate_fn
        self.interval = 0

    def _samples(self, reader: int, n_readers: int, rng: random.Random):
        shards = list(self.shards)
        if self.is_train:
            random.Random(self.seed + 7919 * self.interval).shuffle(shards)      # same order on every reader
        by_shard = len(shards) >= n_readers
        mine = shards[reader::n_readers] if by_shard else shards
        idx = 0
        for sh in mine:
            buf = []
            for raw in iter_shard(sh):
                take = by_shard or (idx % n_readers == reader)
                idx += 1
                if not take:
                    continue
                try:
                    buf.append(self.decoder(raw, rng))
                except Exception as e:  # noqa: BLE001   corrupt page / empty annotation: skip like the reference pipeline
                    _logger.warning(f'skipping sample {raw.get("__key__")}: {type(e).__name__}: {e}')
                    continue
                if self.is_train and len(buf) >= 64:        # local shuffle buffer
                    rng.shuffle(buf)
                    while len(buf) > 32:
                        yield buf.pop()
            if self.is_train:
                rng.shuffle(buf)
            yield from buf

    def __iter__(self):
        wi = get_worker_info()
        nw, wid = (wi.num_workers, wi.id) if wi is not None else (1, 0)
        reader, n_readers = self.global_rank * nw + wid, self.world_size * nw
        rng = random.Random(self.seed + 1000003 * self.interval + reader)
        if not self.is_train:
            # evaluation: every reader walks its part ONCE and emits all of it, the last (partial) batch included -- no quota, so no
            # sample is dropped whatever the worker count (100 samples, batch 8, 4 workers: 4 x (3 full + 1 partial) batches)
            batch = []
            for s in self._samples(reader, n_readers, rng):
                batch.append(s)
                if len(batch) == self.batch_size:
                    yield self.collate_fn(batch)
                    batch = []
            if batch:
                yield self.collate_fn(batch)
            return
        quota = self.num_batches // nw + (1 if wid < self.num_batches % nw else 0)     # batches this worker contributes
        made, batch, passes = 0, [], 0
        while made < quota:
            got = False
            for s in self._samples(reader, n_readers, rng):
                got = True
                batch.append(s)
                if len(batch) == self.batch_size:
                    yield self.collate_fn(batch)
                    batch = []
                    made += 1
                    if made >= quota:
                        return
            passes += 1
            if not got:
                raise RuntimeError(f'reader {reader}: no usable sample in {self.shards}')
            # training: the interval length is num_samples, shards are re-read (reshuffled by the rng) until it is reached


class _IntervalLoader:
    """iterable handed to train_one_interval: DataLoader over the shard reader (+ the on-device image stage)"""

    def __init__(self, dataset: _DocShards, num_workers: int, image_preprocess, device_stage: bool):
        self.dataset, self.image_preprocess, self.device_stage = dataset, image_preprocess, device_stage
        pin = torch.cuda.is_available() and not device_stage
        self.dl = DataLoader(dataset, batch_size=None, num_workers=num_workers, pin_memory=pin,
                             persistent_workers=False, prefetch_factor=2 if num_workers > 0 else None)
        self._pinned = []          # one reusable pinned staging buffer per batch slot (on-device preprocessing): no hipHostMalloc per page
        self._staged = None        # event recorded after the last batch's uploads were enqueued

    def _stage(self, i: int, pg: torch.Tensor) -> torch.Tensor:
        n = pg.numel()
        while len(self._pinned) <= i:
            self._pinned.append(None)
        if self._pinned[i] is None or self._pinned[i].numel() < n:
            self._pinned[i] = torch.empty(int(n * 1.25) + 4096, dtype=torch.uint8).pin_memory()
        view = self._pinned[i][:n].view(pg.shape)
        view.copy_(pg)
        return view

    def set_interval(self, interval: int):
        self.dataset.interval = interval

    def __len__(self):
        return self.dataset.num_batches

    def __iter__(self):
        n = 0
        for batch in self.dl:
            if self.dataset.is_train and n >= self.dataset.num_batches:     # evaluation runs to exhaustion (partial tail batches included)
                break
            n += 1
            if self.device_stage:
                pages = batch[0]
                pre = self.image_preprocess
                out = torch.empty(len(pages), pre.C, pre.Ho, pre.Wo, dtype=torch.float32, device=pre.device)
                if self._staged is not None:
                    self._staged.synchronize()           # the previous batch's uploads have left the staging buffers
                for i, pg in enumerate(pages):           # decoded uint8 page -> HIP resize + normalise, straight into the batch
                    pre(self._stage(i, pg) if pg.device.type == 'cpu' else pg, out=out[i])
                self._staged = torch.cuda.Event()
                self._staged.record()
                batch = (out,) + tuple(batch[1:])
            yield batch


def create_loader(cfg: DatasetCfg, is_train: bool, image_preprocess, anno_preprocess, collate_fn: Callable = None,
                  image_key: str = DEFAULT_IMAGE_KEY, image_fmt: str = 'L', start_interval: int = 0, seed: int = 0,
                  world_size: int = 1, global_rank: int = 0, create_decoder_pipe: Callable = create_doc_anno_pipe) -> LoaderBundle:
    """ref data/loader.py:24-119 (same parameters, same return contract)"""
    if cfg.format == 'webdataset':
        decoder = create_decoder_pipe(image_preprocess=image_preprocess, anno_preprocess=anno_preprocess, image_key=image_key,
                                      image_fmt=image_fmt)
        shards = expand_source(cfg.source)
        for sh in shards:
            if not os.path.exists(sh):
                raise FileNotFoundError(f'shard {sh!r} of source {cfg.source!r} does not exist')
        num_batches = cfg.num_samples // (cfg.batch_size * world_size)
        if not is_train and cfg.num_samples % (cfg.batch_size * world_size):
            num_batches += 1
        assert num_batches > 0, f'num_samples {cfg.num_samples} < one global batch ({cfg.batch_size} x {world_size})'
        wds_collate = partial(collate_doc_pages, is_train=is_train)
        ds = _DocShards(shards, decoder, cfg.batch_size, num_batches, is_train, seed, world_size, global_rank, wds_collate)
        ds.interval = start_interval
        loader = _IntervalLoader(ds, cfg.num_workers, image_preprocess, bool(getattr(image_preprocess, 'on_device', False)))
        return LoaderBundle(loader=loader, num_batches=num_batches, num_samples=num_batches * cfg.batch_size, sampler=None)
    if cfg.format == 'hf_dataset':
        # the task-level collate_fn builds the batch (ref :84-118); local datasets only -- there is no network
        import datasets
        if os.path.isdir(cfg.source):
            ds = datasets.load_from_disk(cfg.source)
            ds = ds[cfg.split] if isinstance(ds, datasets.DatasetDict) else ds
        else:
            ds = datasets.load_dataset(cfg.source)[cfg.split]
        sampler = None
        if world_size > 1:
            sampler = DistributedSampler(ds, rank=global_rank, shuffle=True, seed=seed, num_replicas=world_size, drop_last=True)
        base = DataLoader(dataset=ds, collate_fn=collate_fn, sampler=sampler, batch_size=cfg.batch_size, num_workers=cfg.num_workers)
        return LoaderBundle(loader=base, num_batches=len(base), num_samples=len(ds), sampler=sampler)
    raise ValueError(f'unknown dataset format {cfg.format!r} (webdataset | hf_dataset)')


class DeviceImagePreprocess:
    """`image_preprocess` for create_loader that defers the resize + normalise to the GPU (GpuImagePreprocess built lazily
    in the main process; the worker processes never touch the device)"""
    on_device = True

    def __init__(self, image_size, mean, std, num_chs, device):
        self.args = (tuple(image_size), mean, std, num_chs, device)
        self.Ho, self.Wo, self.C, self.device = int(image_size[0]), int(image_size[1]), num_chs, device
        self._impl = None

    def __getstate__(self):     # workers get the description only
        d = dict(self.__dict__)
        d['_impl'] = None
        return d

    def __call__(self, page_u8: torch.Tensor, out: torch.Tensor = None) -> torch.Tensor:
        if self._impl is None:
            from .gpu_preprocess import GpuImagePreprocess
            self._impl = GpuImagePreprocess(*self.args)
        return self._impl(page_u8, out=out)
