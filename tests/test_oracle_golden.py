"""Pins oracle/ref_cpu.py against the golden vectors (CPU only).

G1/G2: live transformers BartForCausalLM outputs; G3: reference preprocess.py outputs;
G4: HF ViT / CLIP-ViT / Swin outputs (timm stand-ins); G5: torch AdamW / clip / cosine formula.
"""
import json
import os

import pytest
import torch
from safetensors.torch import load_file

from oracle import ref_cpu as R


def _load(golden_dir, tag):
    t = load_file(os.path.join(golden_dir, tag + '.safetensors'))
    meta = json.load(open(os.path.join(golden_dir, tag + '.json')))
    return t, meta


def _decoder_case(golden_dir, tag):
    t, meta = _load(golden_dir, tag)
    arch = dict(d_model=meta['d_model'], heads=meta['heads'], ffn=meta['ffn'], ln_eps=1e-5)
    params = {k[2:]: v.float() for k, v in t.items() if k.startswith('w.')}
    return t, meta, arch, params


@pytest.mark.parametrize('tag', ['g1_decoder_tiny', 'g2_decoder_hd64'])
def test_decoder_fp32_matches_transformers(golden_dir, tag):
    t, meta, arch, params = _decoder_case(golden_dir, tag)
    params = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    enc = t['in.enc'].clone().requires_grad_(True)
    logits = R.bart_decoder_forward(params, arch, meta['layers'], t['in.input_ids'], enc, 'fp32')
    lc = meta['logit_cols']
    assert torch.allclose(logits[:, :, :lc], t['out.logits_fp32'], atol=2e-5, rtol=1e-5)
    loss = R.cross_entropy(logits, t['in.target'])
    assert abs(float(loss) - float(t['out.loss_fp32'])) < 1e-5
    loss.backward()
    assert torch.allclose(enc.grad, t['out.grad_enc'], atol=1e-6, rtol=1e-4)
    for name, gn in meta['grad_norms'].items():
        g = params[name].grad
        assert abs(float(g.norm()) - gn) <= 1e-4 * max(gn, 1e-3), name
        if ('g.' + name) in t:
            assert torch.allclose(g, t['g.' + name], atol=1e-6, rtol=1e-4), name


@pytest.mark.parametrize('tag', ['g1_decoder_tiny', 'g2_decoder_hd64'])
def test_decoder_bf16_policy_close_to_autocast(golden_dir, tag):
    """explicit CUDA-autocast policy vs the CPU-autocast run of the live model: same to bf16 noise."""
    t, meta, arch, params = _decoder_case(golden_dir, tag)
    logits = R.bart_decoder_forward(params, arch, meta['layers'], t['in.input_ids'], t['in.enc'], 'bf16')
    assert logits.dtype == torch.bfloat16
    loss = R.cross_entropy(logits, t['in.target'])
    ref = float(t['out.loss_bf16'])
    assert abs(float(loss) - ref) / ref < 1e-3
    assert abs(float(loss) - float(t['out.loss_fp32'])) / ref < 1e-3
    lc = meta['logit_cols']
    err = (logits[:, :, :lc].float() - t['out.logits_fp32']).abs().max()
    assert err < 0.06 * t['out.logits_fp32'].abs().max()


def test_preprocess_targets(golden_dir):
    cases = json.load(open(os.path.join(golden_dir, 'g3_preprocess.json')))
    assert cases
    for c in cases:
        text = torch.tensor(c['text'])
        tgt = R.make_targets(text, pad_token_id=1, prompt_end_token_id=50266)
        assert tgt.tolist() == c['target'], c['anno']
    # the known answer quoted in SURVEY App. A.4
    c = cases[0]
    assert c['text'][0] == 50266 and c['target'][0] == -100 and c['meta']['page_indices'] == [1]
    ti, tt = R.shift_tokens(torch.tensor([c['text']]), torch.tensor([c['target']]))
    assert ti.shape[1] == c['L'] - 1 and tt[0, 0] != -100


@pytest.mark.parametrize('tag', ['g4_vit', 'g4_clip'])
def test_vit_matches_hf(golden_dir, tag):
    t, arch = _load(golden_dir, tag)
    params = {k[2:]: v for k, v in t.items() if k.startswith('w.')}
    shapes = R.vit_param_shapes(arch, arch['in_chans'], tuple(arch['img_size']))
    assert {k: tuple(v.shape) for k, v in params.items()} == shapes
    out = R.vit_forward(params, arch, t['in.image'], 'fp32')
    assert torch.allclose(out, t['out.tokens'], atol=2e-5, rtol=1e-5)
    out16 = R.vit_forward(params, arch, t['in.image'], 'bf16')
    assert out16.dtype == torch.float32
    assert (out16 - t['out.tokens']).abs().max() < 0.08


@pytest.mark.parametrize('tag', ['g4_swin_shift', 'g4_swin_clamp'])
def test_swin_matches_hf(golden_dir, tag):
    t, arch = _load(golden_dir, tag)
    arch['depths'], arch['heads'] = tuple(arch['depths']), tuple(arch['heads'])
    params = {k[2:]: v for k, v in t.items() if k.startswith('w.')}
    shapes = R.swin_param_shapes(arch, arch['in_chans'], tuple(arch['img_size']))
    assert {k: tuple(v.shape) for k, v in params.items()} == shapes
    out = R.swin_forward(params, arch, t['in.image'], 'fp32')
    assert torch.allclose(out, t['out.tokens'], atol=2e-5, rtol=1e-5)
    out16 = R.swin_forward(params, arch, t['in.image'], 'bf16')
    assert (out16 - t['out.tokens']).abs().max() < 0.1


def test_adamw_clip_schedule(golden_dir):
    t, meta = _load(golden_dir, 'g5_optim')
    n = meta['n_tensors']
    p = [t[f'p0.{i}'].clone() for i in range(n)]
    m = [torch.zeros_like(x) for x in p]
    v = [torch.zeros_like(x) for x in p]
    for step in range(3):
        g = [t[f'g{step}.{i}'].clone() for i in range(n)]
        total = R.clip_grad_norm_(g, meta['clip'])
        assert abs(float(total) - meta['grad_norms'][step]) < 1e-5 * meta['grad_norms'][step]
        for i in range(n):
            assert torch.allclose(g[i], t[f'gclip{step}.{i}'], atol=1e-7, rtol=1e-6)
            R.adamw_step(p[i], g[i], m[i], v[i], step + 1, meta['lr'], meta['betas'][0], meta['betas'][1], meta['eps'])
            assert torch.allclose(p[i], t[f'p{step + 1}.{i}'], atol=1e-7, rtol=1e-6)
    s = meta['sched']
    for k, lr in s['lrs'].items():
        assert abs(R.cosine_lr(int(k), s['base'], s['warmup_t'], s['t_initial']) - lr) < 1e-12


def test_oracle_train_step_runs_cfg1_shape():
    """cfg-1 architecture (swin_tiny + BART-base 2L) at reduced sequence: loss finite, params move."""
    spec = R.ModelSpec('swin_tiny_patch4_window7_224', 'facebook/bart-base', 2, 16, (224, 224), 3, vocab=1027)
    params = R.init_params(spec, seed=0)
    tr = R.OracleTrainer(spec, params, lr=1e-3, betas=(0.9, 0.98), clip_grad=1.0, warmup_t=2, t_initial=10)
    sample = R.synthetic_sample(spec, 1, ragged=True)
    l0 = tr.train_step(sample)
    assert l0 == l0 and tr.step == 1 and tr.last_grad_norm > 0
    assert abs(tr.lr - R.cosine_lr(1, 1e-3, 2, 10)) < 1e-15
    k = 'text_decoder.trunk.model.decoder.layers.0.fc1.weight'
    # warmup_lr_init = 0 => the first update runs at lr 0 (scheduler.step_update(0), task_cruller_pretrain.py:224)
    assert torch.equal(tr.params[k].detach(), params[k])
    tr.train_step(sample)
    assert not torch.equal(tr.params[k].detach(), params[k])
