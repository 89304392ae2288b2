"""End-to-end throughput of the real-data loader boundary (ref data/loader.py:24-119, app/train.py:171-183):
tar shards of A4 page scans (+ .json OCR annotations) -> worker decode (PIL) -> uint8 pages -> pinned staging -> H2D -> HIP
antialiased-bicubic resize + normalise -> device batch [B, 3, 1280, 960] + token tensors -- what TaskCrullerPretrain.train_step
consumes on cfg-3.  Reports docs/s against the worker count, next to the ~31 docs/s one GPU's training step consumes
(8 GPUs: ~250 docs/s from one host).

    python scripts/bench_loader.py [--docs 256] [--workers 0,4,8,16,32] [--fmt png|jpg] [--cpu-resize]

--cpu-resize runs the reference's arrangement instead (the torchvision-equivalent resize in the workers, fp32 batches over PCIe).
Synthetic pages: 1754 x 1240 (A4 @ 150 dpi) with text-like structure so that PNG / JPEG decode costs are realistic."""
import argparse
import io
import json
import os
import sys
import tarfile
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch


def make_page(rng, h=1754, w=1240):
    """white page with dark 'text lines' of random glyph-sized blobs + scanner noise"""
    a = np.full((h, w), 245, np.uint8)
    for y in range(120, h - 120, 34):
        x = 100
        while x < w - 140:
            wl = int(rng.randint(20, 110))
            if rng.rand() < 0.9:
                blk = rng.randint(0, 2, (18, wl)).astype(np.uint8) * 200
                a[y:y + 18, x:x + wl] = np.minimum(a[y:y + 18, x:x + wl], 250 - blk)
            x += wl + int(rng.randint(8, 24))
    a = np.clip(a.astype(np.int16) + rng.randint(-6, 7, a.shape), 0, 255).astype(np.uint8)
    return np.stack([a, a, a], -1)


def build_shards(root, n_docs, fmt, n_shards=8):
    from PIL import Image
    rng = np.random.RandomState(0)
    sizes, encoded = [], []
    for _ in range(8):                                # 8 distinct pages encoded once; every document carries one of them (the DECODE cost is what is measured)
        buf = io.BytesIO()
        Image.fromarray(make_page(rng)).save(buf, format='PNG' if fmt == 'png' else 'JPEG', **({} if fmt == 'png' else {'quality': 90}))
        encoded.append(buf.getvalue())
    for s in range(n_shards):
        with tarfile.open(os.path.join(root, f'shard-{s:03d}.tar'), 'w') as tf:
            for i in range(s, n_docs, n_shards):
                page = encoded[i % len(encoded)]
                js = json.dumps({'pages': [{'text': [f'line {j} of document {i}: lorem ipsum dolor sit amet' for j in range(40)]}]}).encode()
                for ext, data in ((fmt, page), ('json', js)):
                    ti = tarfile.TarInfo(f'doc{i:06d}.{ext}')
                    ti.size = len(data)
                    tf.addfile(ti, io.BytesIO(data))
                sizes.append(len(page))
    return float(np.mean(sizes))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--docs', type=int, default=256)
    ap.add_argument('--workers', default='0,4,8,16,32')
    ap.add_argument('--fmt', default='png', choices=['png', 'jpg'])
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--cpu-resize', action='store_true')
    a = ap.parse_args()
    from functools import partial
    from pixparse_amd.data import DatasetCfg, create_loader, preprocess_ocr_anno
    from pixparse_amd.data.loader import DeviceImagePreprocess
    from pixparse_amd.task.task_cruller_pretrain import ImagePreprocess
    from pixparse_amd.tokenizers import ByteBartTokenizer
    dev = torch.device('cuda:0')
    tok = ByteBartTokenizer()
    tok.add_special_tokens({'additional_special_tokens': ['<s_pretrain>']})
    anno = partial(preprocess_ocr_anno, tokenizer=tok, max_position_embeddings=1024, task_start_token='<s_pretrain>', prompt_end_token='<s_pretrain>')
    size, mean, std = (1280, 960), (0.5, 0.5, 0.5), (0.5, 0.5, 0.5)
    with tempfile.TemporaryDirectory() as root:
        t0 = time.time()
        avg = build_shards(root, a.docs, a.fmt)
        print(f'# {a.docs} pages 1754x1240 RGB as {a.fmt} (mean {avg / 1e3:.0f} KB) in 8 tar shards, built in {time.time() - t0:.0f} s; '
              f'{os.cpu_count()} host cores; image stage: {"CPU resize in the workers (reference arrangement)" if a.cpu_resize else "workers decode only, HIP resize + normalise on the device"}')
        print('# workers  docs/s   ms/batch(8)')
        for nw in [int(x) for x in a.workers.split(',')]:
            pre = ImagePreprocess(size, mean, std, 3) if a.cpu_resize else DeviceImagePreprocess(size, mean, std, 3, dev)
            cfg = DatasetCfg(source=os.path.join(root, 'shard-{000..007}.tar'), num_samples=a.docs, batch_size=a.batch, num_workers=nw)
            b = create_loader(cfg, is_train=True, image_preprocess=pre, anno_preprocess=anno, image_fmt='RGB', seed=1)
            b.set_interval(0)
            n, t0 = 0, None
            for image, text, target in b.loader:
                image = image.to(dev, non_blocking=True)
                text = text.to(dev, non_blocking=True)
                if t0 is None:                      # the first batch pays the worker start-up: timed from the second on
                    torch.cuda.synchronize()
                    t0 = time.time()
                    continue
                n += image.shape[0]
            torch.cuda.synchronize()
            dt = time.time() - t0
            assert image.shape == (a.batch, 3, 1280, 960) and text.shape == (a.batch, 1024)
            print(f'  {nw:7d}  {n / dt:7.1f}  {dt / (n / a.batch) * 1e3:9.1f}', flush=True)


if __name__ == '__main__':
    main()
