"""CPU tests of the real-data loader boundary (ref data/loader.py:24-119, app/train.py:171-183):
tar shards / directories of (page image, .json annotation) -> batches with the train_step contract.  Parity: every sample
the loader emits equals the task's own preprocessing applied by hand -- ImagePreprocess (torch's bicubic-antialias resize
= the torchvision Compose of task_cruller_pretrain.py:132-143) and preprocess_ocr_anno, which tests/golden G3 pins to the
reference's data/preprocess.py."""
import io
import json
import os
import random
import tarfile

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _make_docs(n, seed=0):
    from PIL import Image
    rng = np.random.RandomState(seed)
    docs = []
    for i in range(n):
        h, w = 40 + 3 * (i % 5), 30 + 2 * (i % 7)
        img = Image.fromarray(rng.randint(0, 256, (h, w), dtype=np.uint8), mode='L')
        buf = io.BytesIO()
        img.save(buf, format='PNG')
        pages = [{'text': [f'doc {i} line {j}' for j in range(1 + i % 3)]}]
        if i % 4 == 0:
            pages.insert(0, {'text': []})            # an empty first page: get_next_valid_page_index must skip it
        docs.append((f'doc{i:04d}', buf.getvalue(), json.dumps({'pages': pages}).encode()))
    return docs


def _write_tar(path, docs):
    with tarfile.open(path, 'w') as tf:
        for key, png, js in docs:
            for ext, data in (('png', png), ('json', js)):
                ti = tarfile.TarInfo(f'{key}.{ext}')
                ti.size = len(data)
                tf.addfile(ti, io.BytesIO(data))


def _write_dir(path, docs):
    os.makedirs(path, exist_ok=True)
    for key, png, js in docs:
        open(os.path.join(path, key + '.png'), 'wb').write(png)
        open(os.path.join(path, key + '.json'), 'wb').write(js)


def _task_fns(L=32, size=(24, 16)):
    from functools import partial
    from pixparse_amd.data import preprocess_ocr_anno
    from pixparse_amd.task.task_cruller_pretrain import ImagePreprocess
    from pixparse_amd.tokenizers import ByteBartTokenizer
    tok = ByteBartTokenizer()
    tok.add_special_tokens({'additional_special_tokens': ['<s_pretrain>']})
    anno = partial(preprocess_ocr_anno, tokenizer=tok, max_position_embeddings=L, task_start_token='<s_pretrain>', prompt_end_token='<s_pretrain>')
    return ImagePreprocess(size, 0.5, 0.25, 1), anno, tok


def test_webdataset_loader_contract_and_sample_parity(tmp_path):
    from PIL import Image
    from pixparse_amd.data import DatasetCfg, LoaderBundle, create_loader
    docs = _make_docs(23)
    for s in range(3):
        _write_tar(str(tmp_path / f'shard-{s:03d}.tar'), docs[s::3])
    img_pre, anno_pre, tok = _task_fns()
    cfg = DatasetCfg(source=str(tmp_path / 'shard-{000..002}.tar'), num_samples=20, batch_size=4, num_workers=0)
    b = create_loader(cfg, is_train=True, image_preprocess=img_pre, anno_preprocess=anno_pre, image_fmt='L', seed=3)
    assert isinstance(b, LoaderBundle) and b.num_batches == 5 and b.num_samples == 20 and b.sampler is None
    b.set_interval(0)
    batches = list(b.loader)
    assert len(batches) == 5
    by_key = {k: (png, js) for k, png, js in docs}
    seen = 0
    for image, text, target in batches:
        assert image.shape == (4, 1, 24, 16) and image.dtype == torch.float32
        assert text.shape == (4, 32) and text.dtype == torch.int64 and target.shape == (4, 32)
        for i in range(4):
            # which document is this? the text decodes back to its lines
            s = tok.decode(text[i].tolist(), skip_special_tokens=True)
            idx = int(s.split()[1])
            png, js = by_key[f'doc{idx:04d}']
            want_img = img_pre(Image.open(io.BytesIO(png)).convert('L'))
            assert torch.equal(image[i], want_img)
            want, meta = anno_pre(json.loads(js), generator=random.Random(0))
            assert torch.equal(text[i], want['text'][0]) and torch.equal(target[i], want['target'][0])
            assert (target[i] == -100).sum() >= 1 and int(text[i, 0]) == 50265        # <s_pretrain> first, masked in the target
            seen += 1
    assert seen == 20
    # reshuffled per interval, reproducible per (seed, interval)
    b.set_interval(1)
    again1 = [t for _, t, _ in b.loader]
    b.set_interval(1)
    again2 = [t for _, t, _ in b.loader]
    assert all(torch.equal(x, y) for x, y in zip(again1, again2))
    assert not all(torch.equal(x, y[1]) for x, y in zip(again1, batches))


def test_directory_source_equals_tar_source_and_eval_format(tmp_path):
    from pixparse_amd.data import DatasetCfg, create_loader
    docs = _make_docs(9, seed=1)
    _write_tar(str(tmp_path / 'a.tar'), docs)
    _write_dir(str(tmp_path / 'adir'), docs)
    img_pre, anno_pre, _ = _task_fns()
    out = []
    for src in ('a.tar', 'adir'):
        cfg = DatasetCfg(source=str(tmp_path / src), num_samples=9, batch_size=4, num_workers=0)
        b = create_loader(cfg, is_train=False, image_preprocess=img_pre, anno_preprocess=anno_pre, image_fmt='L')
        assert b.num_batches == 3
        out.append(list(b.loader))
    for (ia, ta, ga), (ib, tb, gb) in zip(*out):
        assert torch.equal(ia, ib)
        # eval batches: per-document lists of per-page tensors (what task_cruller_eval_ocr.step indexes with item[0])
        assert isinstance(ta, list) and isinstance(ta[0], list) and ta[0][0].shape == (32,)
        assert all(torch.equal(x[0], y[0]) for x, y in zip(ta, tb)) and all(torch.equal(x[0], y[0]) for x, y in zip(ga, gb))
    assert [len(t) for _, t, _ in out[0]] == [4, 4, 1]                     # eval keeps the ragged last batch


def test_eval_with_workers_sees_every_sample_once(tmp_path):
    """evaluation with several workers: every reader emits ALL of its part, the partial tail batch included (no per-worker batch
    quota): 23 documents, batch 4, 3 workers -> all 23 documents, each exactly once"""
    from pixparse_amd.data import DatasetCfg, create_loader
    docs = _make_docs(23, seed=4)
    _write_tar(str(tmp_path / 'e.tar'), docs)
    img_pre, anno_pre, tok = _task_fns()
    cfg = DatasetCfg(source=str(tmp_path / 'e.tar'), num_samples=23, batch_size=4, num_workers=3)
    b = create_loader(cfg, is_train=False, image_preprocess=img_pre, anno_preprocess=anno_pre, image_fmt='L')
    ids = []
    for image, text, _ in b.loader:
        assert image.shape[0] == len(text) <= 4
        ids += [int(tok.decode(t[0].tolist(), skip_special_tokens=True).split()[1]) for t in text]
    assert sorted(ids) == list(range(23))


def test_ranks_and_workers_read_disjoint_samples(tmp_path):
    from pixparse_amd.data import DatasetCfg, create_loader
    docs = _make_docs(32, seed=2)
    for s in range(4):
        _write_tar(str(tmp_path / f's{s}.tar'), docs[s * 8:(s + 1) * 8])
    img_pre, anno_pre, tok = _task_fns()
    seen = []
    for rank in range(2):
        cfg = DatasetCfg(source=str(tmp_path / 's*.tar'), num_samples=32, batch_size=4, num_workers=2)
        b = create_loader(cfg, is_train=True, image_preprocess=img_pre, anno_preprocess=anno_pre, image_fmt='L', seed=5, world_size=2, global_rank=rank)
        assert b.num_batches == 4
        ids = []
        for _, text, _ in b.loader:
            ids += [int(tok.decode(t.tolist(), skip_special_tokens=True).split()[1]) for t in text]
        assert len(ids) == 16 and len(set(ids)) == 16
        seen.append(set(ids))
    assert not (seen[0] & seen[1]) and (seen[0] | seen[1]) == set(range(32))


def test_bad_samples_are_skipped_and_errors_are_loud(tmp_path):
    from pixparse_amd.data import DatasetCfg, create_loader
    docs = _make_docs(6, seed=3)
    docs[1] = (docs[1][0], b'not a png', docs[1][2])                       # corrupt page
    docs[2] = (docs[2][0], docs[2][1], json.dumps({'pages': []}).encode())   # empty annotation (reference raises -> skipped)
    _write_tar(str(tmp_path / 'x.tar'), docs)
    img_pre, anno_pre, _ = _task_fns()
    cfg = DatasetCfg(source=str(tmp_path / 'x.tar'), num_samples=4, batch_size=4, num_workers=0)
    b = create_loader(cfg, is_train=False, image_preprocess=img_pre, anno_preprocess=anno_pre, image_fmt='L')
    (image, text, _), = list(b.loader)
    assert image.shape[0] == 4 and len(text) == 4
    with pytest.raises(FileNotFoundError):
        create_loader(DatasetCfg(source=str(tmp_path / 'missing.tar'), num_samples=4, batch_size=4), True, img_pre, anno_pre)
    with pytest.raises(ValueError, match='unknown dataset format'):
        create_loader(DatasetCfg(source='x', num_samples=4, batch_size=4, format='csv'), True, img_pre, anno_pre)
