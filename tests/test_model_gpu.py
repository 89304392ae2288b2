"""Model-level parity on the MI355X: HIP engines vs the CPU oracle / golden vectors (`-m gpu`).

Tolerances: the HIP path and the oracle both implement the bf16 autocast policy but round at
different points inside fused kernels, so activations agree to a few bf16 ulps (2^-8 relative);
the LOSS must agree to 1e-3 relative (BASELINE.json north_star) -- asserted below.
"""
import json
import os

import math

import pytest
import torch
from safetensors.torch import load_file

pytestmark = pytest.mark.gpu
BF16, F32 = torch.bfloat16, torch.float32


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available()
    return torch.device('cuda:0')


def rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-12))


def _register_test_archs():
    from pixparse_amd.models.archs import register_arch
    from oracle import ref_cpu as R
    vit_a = dict(patch=8, dim=128, depth=2, heads=2, mlp_ratio=4, ln_eps=1e-6, pre_norm=False, mean=(0.5,) * 3, std=(0.5,) * 3)
    vit_b = dict(vit_a, pre_norm=True, ln_eps=1e-5)
    swin = dict(patch=4, embed_dim=32, depths=(2, 2, 2), heads=(1, 2, 4), window=4, mlp_ratio=4, ln_eps=1e-5, mean=(0.5,) * 3, std=(0.5,) * 3)
    bart = dict(d_model=128, heads=2, ffn=256, ln_eps=1e-5, vocab=509, dropout=0.0)
    for reg, Rv, Rs, Rb in ((register_arch, R.VIT_ARCHS, R.SWIN_ARCHS, R.BART_ARCHS),):
        reg('vit', 'vit_test', vit_a); reg('vit', 'vit_test_clip', vit_b); reg('swin', 'swin_test', swin); reg('bart', 'bart_test', bart)
        Rv['vit_test'] = vit_a; Rv['vit_test_clip'] = vit_b; Rs['swin_test'] = swin; Rb['bart_test'] = bart
        vit_768 = dict(vit_a, patch=16, dim=768, depth=1, heads=12)          # the width the reference's classifier head hard-codes
        bart_768 = dict(bart, d_model=768, heads=12)
        reg('vit', 'vit_test768', vit_768); reg('bart', 'bart_test768', bart_768)
        Rv['vit_test768'] = vit_768; Rb['bart_test768'] = bart_768


def _cfg(enc, img, fmt, layers, L):
    from pixparse_amd.models import ImageEncoderCfg, ModelCfg, TextDecoderCfg
    return ModelCfg(image_encoder=ImageEncoderCfg(name=enc, image_fmt=fmt, image_size=img, pretrained=False),
                    text_decoder=TextDecoderCfg(name='bart_test', pretrained=False, num_decoder_layers=layers, max_length=L))


def test_decoder_engine_vs_transformers_golden(dev, golden_dir):
    """G2: live BartForCausalLM numbers (head_dim 64, odd vocab 1027, ragged targets)."""
    from pixparse_amd import ops
    from pixparse_amd.layers.arena import ParamArena
    from pixparse_amd.layers.engines import BartEngine, Buffers
    t = load_file(os.path.join(golden_dir, 'g2_decoder_hd64.safetensors'))
    meta = json.load(open(os.path.join(golden_dir, 'g2_decoder_hd64.json')))
    arch = dict(d_model=meta['d_model'], heads=meta['heads'], ffn=meta['ffn'], ln_eps=1e-5)
    B, T, S, V = meta['B'], meta['T'], meta['S'], meta['vocab']
    arena = ParamArena()
    for item in BartEngine.param_shapes(arch, meta['layers'], V, T + 1):
        arena.add(item[0], item[1], item[2] if len(item) > 2 else None)
    arena.materialize(dev)
    for k, v in t.items():
        if k.startswith('w.'):
            arena.param(k[2:]).copy_(v.float())
    arena.alloc_training_state()
    arena.alloc_shadow()
    ops.cast_bf16(arena.p, arena.pb)
    eng = BartEngine(arch, meta['layers'], V, T + 1, arena, '', Buffers(dev))
    enc16 = t['in.enc'].to(dev).to(BF16).view(B * S, -1).contiguous()
    ids, target = t['in.input_ids'].to(dev), t['in.target'].to(dev)
    logits = eng.forward(ids, enc16, S)
    lc = meta['logit_cols']
    got = logits.view(B, T, eng.Vp)[:, :, :lc].float().cpu()
    ref = t['out.logits_fp32']
    assert (got - ref).abs().max() < 0.03 * ref.abs().max(), 'logits vs fp32 transformers'
    loss = torch.zeros(1, device=dev); nv = torch.zeros(1, dtype=torch.int32, device=dev); rl = torch.empty(B * T, device=dev)
    ops.cross_entropy(logits, target.view(-1), V, 1.0, 1.0, loss, nv, rl, logits)
    assert abs(float(loss) - float(t['out.loss_fp32'])) / float(t['out.loss_fp32']) < 1e-3
    denc = torch.zeros(B * S, arch['d_model'], device=dev)
    eng.backward(logits, enc16, denc)
    assert rel(denc.view(B, S, -1), t['out.grad_enc']) < 3e-2
    for name, gn in meta['grad_norms'].items():
        g = arena.grad(name)
        if name.endswith('k_proj.bias'):
            # d loss / d k-bias is identically zero (a key bias shifts every score of a row equally and softmax is
            # shift invariant): both sides hold only rounding noise, so compare against the q-bias scale instead
            assert float(g.norm()) < 5e-2 * meta['grad_norms'][name.replace('k_proj', 'q_proj')], name
            continue
        assert abs(float(g.norm()) - gn) <= 3e-2 * gn + 1e-6, (name, float(g.norm()), gn)


@pytest.mark.parametrize('enc,img,fmt', [('vit_test', (37, 50), 'RGB'), ('vit_test_clip', (64, 48), 'L'), ('swin_test', (64, 64), 'RGB')])
def test_cruller_forward_backward_vs_oracle(dev, enc, img, fmt):
    from oracle import ref_cpu as R
    from pixparse_amd.models import Cruller
    _register_test_archs()
    L, layers, V, B = 24, 2, 515, 2
    torch.manual_seed(0)
    model = Cruller(_cfg(enc, img, fmt, layers, L), vocab_size=V)
    with torch.no_grad():   # livelier statistics than the 0.02 init: biases and LN offsets non-zero
        for n, p in model.named_parameters():
            if n.endswith('.bias'):
                p.normal_(0, 0.05)
            elif p.dim() >= 2:
                p.mul_(3.0)
    params = {k: v.detach().clone() for k, v in model.state_dict().items() if not k.endswith('lm_head.weight')}
    spec = R.ModelSpec(enc, 'bart_test', layers, L, img, 1 if fmt == 'L' else 3, vocab=V)
    assert {k: tuple(v.shape) for k, v in params.items()} == spec.param_shapes()
    image, tokens, target = R.synthetic_sample(spec, B, seed=3, ragged=True)
    ti, tt = R.shift_tokens(tokens, target)
    # oracle (CPU, bf16 policy) with autograd
    op = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    ologits = R.cruller_forward(op, spec, image, ti, 'bf16')
    oloss = R.cross_entropy(ologits, tt)
    oloss.backward()
    # HIP
    model.to(dev)
    model.arena.alloc_training_state()
    out = model(image.to(dev), ti.to(dev))
    assert out['logits'].shape == (B, L - 1, V) and out.logits.dtype == BF16
    assert rel(out['logits'], ologits) < 2e-2
    loss = model.forward_loss(image.to(dev), ti.to(dev), tt.to(dev))
    assert abs(float(loss) - float(oloss)) / float(oloss) < 1e-3, (float(loss), float(oloss))
    model.backward()
    worst = []
    for k in params:
        g, og = model.arena.grad(k), op[k].grad
        if k.endswith('k_proj.bias'):   # mathematically zero gradient (softmax shift invariance): noise on both sides
            assert float(g.norm()) < 5e-2 * float(op[k.replace('k_proj', 'q_proj')].grad.norm()), k
            continue
        worst.append((rel(g, og), k))
    worst.sort(reverse=True)
    # gradients are sums of bf16-rounded products: 5% relative L2 per tensor, 2% on the total norm
    assert worst[0][0] < 5e-2, worst[:5]
    tot = torch.sqrt(sum((op[k].grad.float() ** 2).sum() for k in params))
    assert abs(float(model.arena.g.norm()) - float(tot)) / float(tot) < 2e-2


def test_classifier_head_forward_backward_vs_oracle(dev):
    """cruller_finetune_xent's model: image encoder -> token 0 -> Linear -> CrossEntropyLoss (ref task_cruller_finetune_xent.py:146-151,
    :237-247) against the oracle's autograd: logits, loss (1e-3), head and encoder gradients; the decoder receives none"""
    from oracle import ref_cpu as R
    from pixparse_amd.models import Cruller
    _register_test_archs()
    enc, img, L, layers, V, B, NC = 'vit_test', (37, 50), 24, 1, 515, 32, 16
    torch.manual_seed(0)
    model = Cruller(_cfg(enc, img, 'RGB', layers, L), vocab_size=V)
    model.add_classifier_head(NC, seed=1)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.endswith('.bias'):
                p.normal_(0, 0.05)
            elif p.dim() >= 2:
                p.mul_(3.0)
    params = {k: v.detach().clone() for k, v in model.state_dict().items() if not k.endswith('lm_head.weight')}
    assert tuple(params['final_fc.weight'].shape) == (NC, 128) and tuple(params['final_fc.bias'].shape) == (NC,)
    spec = R.ModelSpec(enc, 'bart_test', layers, L, img, 3, vocab=V)
    g = torch.Generator().manual_seed(5)
    image = torch.randn(B, 3, *img, generator=g)
    label = torch.randint(0, NC, (B,), generator=g)
    label[1] = -100                                                   # CrossEntropyLoss(ignore_index=-100)
    op = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    oloss, ologits = R.classifier_loss(op, spec, image, label, 'bf16')
    oloss.backward()
    model.to(dev)
    model.arena.alloc_training_state()
    logits = model.classify(image.to(dev))
    assert logits.shape == (B, 32) and rel(logits[:, :NC], ologits) < 2e-2
    loss = model.classify_loss(image.to(dev), label.to(dev))
    # the mean over 31 rows of 16 bf16 logits each (one bf16 ulp of a logit ~ 4e-3): 3e-3, not the 1e-3 of the 10^4-token LM losses
    assert abs(float(loss) - float(oloss)) / float(oloss) < 3e-3, (float(loss), float(oloss))
    model.classify_backward()
    worst = sorted(((rel(model.arena.grad(k), op[k].grad), k) for k in params if op[k].grad is not None
                    and not k.endswith('k_proj.bias') and 'attn.qkv.bias' not in k), reverse=True)
    assert worst[0][0] < 5e-2, worst[:5]
    for k in params:
        if k.startswith('text_decoder.'):
            assert op[k].grad is None and float(model.arena.grad(k).abs().max()) == 0.0, k
    # the padded classifier rows never receive a gradient
    assert float(model.arena.grad('final_fc.weight', padded=True)[NC * 128:].abs().max()) == 0.0


def test_finetune_xent_task_steps(dev, tmp_path):
    """the registry entry: checkpoint handed over as the reference's app/train.py does (task.state_dict = ...; task.resume = True), three
    updates on dict samples, loss falls on a fixed batch, decoder untouched, checkpoint keys of the reference's nn.Sequential"""
    from pixparse_amd.framework import DeviceEnv, Monitor
    from pixparse_amd.models import Cruller
    from pixparse_amd.task import TaskFactory
    _register_test_archs()
    from pixparse_amd.models import ImageEncoderCfg, ModelCfg, TextDecoderCfg
    mcfg = ModelCfg(image_encoder=ImageEncoderCfg(name='vit_test768', image_fmt='RGB', image_size=(32, 48), pretrained=False),
                    text_decoder=TextDecoderCfg(name='bart_test768', pretrained=False, num_decoder_layers=1, max_length=16))
    from pixparse_amd.framework.config import OptimizationCfg
    args = dict(model=mcfg, opt=OptimizationCfg(learning_rate=1e-3, warmup_learning_rate=1e-3, clip_grad_value=1.0), num_intervals=1, num_warmup_intervals=0)
    task, cfg = TaskFactory.create_task('cruller_finetune_xent', dict(args, dtype='bfloat16'), DeviceEnv(), Monitor(output_dir=str(tmp_path)))
    pre = Cruller(mcfg, vocab_size=task.vocab_size)
    ckpt = {'module.' + k: v.clone() for k, v in pre.state_dict().items()}
    task.state_dict = ckpt
    task.resume = True
    task.train_setup(num_batches_per_interval=8)
    assert callable(task.state_dict)
    for k, v in pre.state_dict().items():
        assert torch.equal(task.model.state_dict()[k].cpu(), v), k
    g = torch.Generator().manual_seed(0)
    sample = {'image': torch.randn(4, 3, 32, 48, generator=g), 'label': torch.tensor([3, 0, 15, 7])}
    dec_before = {k: v.clone() for k, v in task.model.state_dict().items() if k.startswith('text_decoder.')}
    task.train_interval_start()
    losses = []
    for _ in range(4):
        task.train_step(sample)
        losses.append(float(task.last_loss))
    assert all(l == l for l in losses) and losses[-1] < losses[0], losses
    assert abs(losses[0] - 2.77) < 0.6                                  # ~ ln 16 at initialisation
    for k, v in dec_before.items():
        assert torch.equal(task.model.state_dict()[k], v), k            # zero gradients, weight decay 0: the decoder does not move
    sd = task.state_dict()['model']
    assert 'final_fc.weight' in sd and tuple(sd['final_fc.weight'].shape) == (16, 768) and 'encoder.trunk.cls_token' in sd
    assert not any(k.startswith('text_decoder') for k in sd)
    assert task.classifier(sample['image'].to(dev)).shape == (4, 16)
    batch = task.collate_fn([{'image': torch.rand(3, 40, 40), 'label': 2}, {'image': torch.rand(3, 50, 30), 'label': 5}])
    assert batch['image'].shape == (2, 3, 32, 48) and batch['label'].tolist() == [2, 5]


def test_decoder_dropout_matches_oracle_with_the_same_masks(dev):
    """hidden-state dropout of the decoder (opt-in).  The oracle applies, at the reference's four dropout sites, the very
    Philox masks the HIP kernels generate (crl_dropout_mask with the same seed / step / site): loss and every gradient then agree to
    the tolerances of the dropout-free test; a second micro-step uses fresh masks; switching it off restores the dropout-free loss."""
    from oracle import ref_cpu as R
    from pixparse_amd import ops
    from pixparse_amd.models import Cruller
    _register_test_archs()
    enc, img, fmt = 'vit_test', (37, 50), 'RGB'
    L, layers, V, B = 24, 2, 515, 2
    torch.manual_seed(0)
    model = Cruller(_cfg(enc, img, fmt, layers, L), vocab_size=V)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.endswith('.bias'):
                p.normal_(0, 0.05)
            elif p.dim() >= 2:
                p.mul_(3.0)
    params = {k: v.detach().clone() for k, v in model.state_dict().items() if not k.endswith('lm_head.weight')}
    spec = R.ModelSpec(enc, 'bart_test', layers, L, img, 1 if fmt == 'L' else 3, vocab=V)
    image, tokens, target = R.synthetic_sample(spec, B, seed=3, ragged=True)
    ti, tt = R.shift_tokens(tokens, target)
    T, D = ti.shape[1], R.BART_ARCHS['bart_test']['d_model']
    p_drop, seed = 0.1, 77
    model.dec_arch = dict(model.dec_arch, dropout=p_drop)
    model.to(dev)
    model.arena.alloc_training_state()
    plain = float(model.forward_loss(image.to(dev), ti.to(dev), tt.to(dev)))
    model.set_train_dropout(True, seed=seed)
    losses = []
    for step in range(2):
        def drop(site, t, step=step):
            keep = ops.dropout_mask(B * T * D, ops.DropSpec(p_drop, seed, step), site, dev).cpu().view(B, T, D).bool()
            return torch.where(keep, (t.float() * (1.0 / (1.0 - p_drop))).to(t.dtype), torch.zeros_like(t))
        op = {k: v.clone().requires_grad_(True) for k, v in params.items()}
        oloss = R.cruller_loss(op, spec, image, ti, tt, 'bf16', drop=drop)
        oloss.backward()
        model.arena.g.zero_()
        loss = float(model.forward_loss(image.to(dev), ti.to(dev), tt.to(dev)))
        model.backward()
        losses.append(loss)
        assert abs(loss - float(oloss)) / float(oloss) < 1e-3, (step, loss, float(oloss))
        worst = sorted(((rel(model.arena.grad(k), op[k].grad), k) for k in params if not k.endswith('k_proj.bias')), reverse=True)
        assert worst[0][0] < 5e-2, worst[:5]
        tot = torch.sqrt(sum((op[k].grad.float() ** 2).sum() for k in params))
        assert abs(float(model.arena.g.norm()) - float(tot)) / float(tot) < 2e-2
    assert abs(losses[0] - plain) > 1e-4 * plain and abs(losses[0] - losses[1]) > 1e-5 * plain     # masks are live and change per micro-step
    # the mask lives for ONE forward_loss() / backward() pair: the reference-style decoder call (ocr_utils) right after a training
    # step -- dropout still switched on -- sees dropout-free logits, also when backward() was never called for that step
    enc_states = model.image_encoder(image.to(dev))
    clean = model(image.to(dev), ti.to(dev)).logits.float().clone()
    after_bwd = model.text_decoder(ti.to(dev), enc_states).logits.float().clone()
    model.forward_loss(image.to(dev), ti.to(dev), tt.to(dev))        # sets a mask, no backward
    no_bwd = model.text_decoder(ti.to(dev), enc_states).logits.float()
    assert torch.equal(after_bwd, clean) and torch.equal(no_bwd, clean)
    model.set_train_dropout(False)
    assert float(model.forward_loss(image.to(dev), ti.to(dev), tt.to(dev))) == plain
    out = model(image.to(dev), ti.to(dev))        # forward() alone (eval / generation paths) never drops
    assert out.logits.shape == (B, T, V)


def test_attention_activation_dropout_and_drop_path_match_oracle(dev):
    """SURVEY K20 completion: the train-mode regularisers of cfg-1's reference models -- bart-base's attention-probability dropout (inside
    the flash kernels: softmax, mask, P.V; the backward passes regenerate the mask) and activation dropout behind the GELU, and the Swin
    encoder's drop-path on both residual branches (per-sample scales, linearly increasing rate) -- next to the hidden-state dropout.
    The oracle applies, at the reference's sites, the very masks / scales the HIP kernels generate (crl_attn_dropout_mask,
    crl_dropout_mask, crl_droppath_scale): loss and every gradient agree to the tolerances of the dropout-free test; switching it off
    restores the dropout-free loss bit for bit; forward() alone never drops."""
    from oracle import ref_cpu as R
    from pixparse_amd import ops
    from pixparse_amd.models import Cruller
    _register_test_archs()
    enc, img, fmt = 'swin_test', (64, 64), 'RGB'
    L, layers, V, B = 24, 2, 515, 2
    torch.manual_seed(1)
    model = Cruller(_cfg(enc, img, fmt, layers, L), vocab_size=V)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.endswith('.bias'):
                p.normal_(0, 0.05)
            elif p.dim() >= 2:
                p.mul_(3.0)
    params = {k: v.detach().clone() for k, v in model.state_dict().items() if not k.endswith('lm_head.weight')}
    spec = R.ModelSpec(enc, 'bart_test', layers, L, img, 3, vocab=V)
    image, tokens, target = R.synthetic_sample(spec, B, seed=4, ragged=True)
    ti, tt = R.shift_tokens(tokens, target)
    T, D = ti.shape[1], R.BART_ARCHS['bart_test']['d_model']
    F_, H = R.BART_ARCHS['bart_test']['ffn'], R.BART_ARCHS['bart_test']['heads']
    p_h, p_a, p_c, p_path, seed = 0.1, 0.15, 0.2, 0.3, 91
    model.dec_arch = dict(model.dec_arch, dropout=p_h, attention_dropout=p_a, activation_dropout=p_c)
    model.enc_arch = dict(model.enc_arch, drop_path=p_path)
    model.to(dev)
    model.arena.alloc_training_state()
    plain = float(model.forward_loss(image.to(dev), ti.to(dev), tt.to(dev)))
    model.set_train_dropout(True, seed=seed)
    e_, d_, _ = model._ensure_engines()
    assert model._drop == (p_h, seed, p_a, p_c, p_path) and e_.nblocks == sum(R.SWIN_ARCHS[enc]['depths'])
    losses = []
    for step in range(2):
        spec_d = ops.DropSpec(p_h, seed, step, p_a, p_c, p_path)

        def hidden(site, t):
            keep = ops.dropout_mask(t.numel(), spec_d, site, dev).cpu().view(t.shape).bool()
            return torch.where(keep, (t.float() * (1.0 / (1.0 - p_h))).to(t.dtype), torch.zeros_like(t))

        def act(site, t):
            keep = ops.dropout_mask(t.numel(), spec_d, site, dev, p=p_c).cpu().view(t.shape).bool()
            return torch.where(keep, (t.float() * (1.0 / (1.0 - p_c))).to(t.dtype), torch.zeros_like(t))

        def attn(site, pr):
            keep = ops.attn_dropout_mask(pr.shape[0], pr.shape[1], pr.shape[2], pr.shape[3], spec_d, site, dev).cpu().bool()
            return torch.where(keep, pr * (1.0 / (1.0 - p_a)), torch.zeros_like(pr))

        def path(site, t):
            j = site // 2
            rate = p_path * j / (e_.nblocks - 1)
            if rate <= 0:
                return t
            sc = torch.empty(B, device=dev)
            ops.droppath_scale(sc, rate, spec_d, site)
            return (t.float() * sc.cpu().view(B, *([1] * (t.dim() - 1)))).to(t.dtype)
        hidden.attn, hidden.act, hidden.path = attn, act, path
        op = {k: v.clone().requires_grad_(True) for k, v in params.items()}
        oloss = R.cruller_loss(op, spec, image, ti, tt, 'bf16', drop=hidden)
        oloss.backward()
        model.arena.g.zero_()
        loss = float(model.forward_loss(image.to(dev), ti.to(dev), tt.to(dev)))
        model.backward()
        losses.append(loss)
        assert abs(loss - float(oloss)) / float(oloss) < 1e-3, (step, loss, float(oloss))
        worst = sorted(((rel(model.arena.grad(k), op[k].grad), k) for k in params if not k.endswith('k_proj.bias')), reverse=True)
        assert worst[0][0] < 5e-2, worst[:5]
        tot = torch.sqrt(sum((op[k].grad.float() ** 2).sum() for k in params))
        assert abs(float(model.arena.g.norm()) - float(tot)) / float(tot) < 2e-2
    assert abs(losses[0] - plain) > 1e-4 * plain and abs(losses[0] - losses[1]) > 1e-5 * plain
    keep = ops.attn_dropout_mask(2, 2, 64, 96, ops.DropSpec(0, seed, 0, p_attn=0.25), 200, dev).float().mean()
    assert abs(float(keep) - 0.75) < 0.02                               # the hash keeps 1 - p of the elements
    model.set_train_dropout(False)
    assert float(model.forward_loss(image.to(dev), ti.to(dev), tt.to(dev))) == plain
    model.set_train_dropout(True, seed=seed)
    model.forward_loss(image.to(dev), ti.to(dev), tt.to(dev))          # masks set, no backward ...
    clean = model(image.to(dev), ti.to(dev)).logits.float().clone()     # ... forward() and the eval paths still see none of them
    model.set_train_dropout(False)
    assert torch.equal(model(image.to(dev), ti.to(dev)).logits.float(), clean)


def test_task_train_steps_vs_oracle_trainer(dev):
    """3 optimiser updates with clip-norm + warmup cosine LR + grad accumulation 2: loss trajectory vs the oracle."""
    from oracle import ref_cpu as R
    from pixparse_amd.framework import DeviceEnv, OptimizationCfg
    from pixparse_amd.task import TaskCrullerPretrain, TaskCrullerPretrainCfg
    _register_test_archs()
    L, layers, img = 24, 2, (37, 50)
    cfg = TaskCrullerPretrainCfg(num_intervals=2, num_warmup_intervals=1, eval_frequency=1000, dtype='bfloat16',
                                 opt=OptimizationCfg(learning_rate=1e-3, betas=(0.9, 0.98), clip_grad_value=1.0, clip_grad_mode='norm',
                                                     grad_accum_steps=2),
                                 model=_cfg('vit_test', img, 'RGB', layers, L))
    torch.manual_seed(1)
    task = TaskCrullerPretrain(cfg, DeviceEnv())
    V = task.vocab_size
    assert V == 50267                                       # 50265 + 2 added tokens
    params = {k: v.detach().clone() for k, v in task.model.state_dict().items() if not k.endswith('lm_head.weight')}
    spec = R.ModelSpec('vit_test', 'bart_test', layers, L, img, 3, vocab=V)
    task.train_setup(num_batches_per_interval=4)            # 2 updates per interval, warmup 2, total 4
    tr = R.OracleTrainer(spec, params, lr=1e-3, betas=(0.9, 0.98), eps=1e-6, clip_grad=1.0, accum_steps=2, warmup_t=2, t_initial=4)
    task.train_interval_start()
    for i in range(6):
        sample = R.synthetic_sample(spec, 2, seed=10 + i, ragged=(i % 2 == 0))
        lo = tr.train_step(sample)
        task.train_step(sample)
        lh = float(task.last_loss)
        assert abs(lh - lo) / abs(lo) < 2e-3, (i, lh, lo)
        assert task.step == tr.step and abs(task.get_current_lr() - tr.lr) < 1e-12
        if (i + 1) % 2 == 0:
            gn = float(task.optimizer.grad_norm())
            assert abs(gn - tr.last_grad_norm) / tr.last_grad_norm < 3e-2, (gn, tr.last_grad_norm)
    assert task.step == 3 and task.batch_idx == 6
    k = 'text_decoder.trunk.model.decoder.layers.0.fc1.weight'
    assert rel(task.model.state_dict()[k], tr.params[k].detach()) < 1e-3
    sd = task.state_dict()
    assert set(sd) == {'model', 'optimizer', 'scheduler', 'scaler'}


def test_app_train_checkpoint_roundtrip(dev, tmp_path):
    """python -m pixparse_amd.app.train on cfg-1 (cruller_small: swin_tiny + BART-base 2L, 224x224, 128 tokens, batch 2):
    two intervals, checkpoint-{i}.pt = model.state_dict() with reference key names, reload (with a DDP 'module.' prefix)"""
    from pixparse_amd.app.train import main
    from pixparse_amd.models import Cruller, get_model_config
    out = str(tmp_path)
    main(['--task.model-name', 'cruller_small', '--task.dtype', 'bfloat16', '--task.opt.learning-rate', '1e-3', '--task.opt.betas', '0.9 0.98',
          '--task.opt.clip-grad-value', '1.0', '--task.opt.clip-grad-mode', 'norm', '--task.num-warmup-intervals', '1',
          '--data.train.batch-size', '2', '--data.train.num-batches', '3', '--train.num-intervals', '2', '--train.output-dir', out,
          '--train.experiment', 'exp'])
    ck = os.path.join(out, 'exp', 'checkpoints')
    assert sorted(os.listdir(ck)) == ['checkpoint-0.pt', 'checkpoint-1.pt']
    sd0 = torch.load(os.path.join(ck, 'checkpoint-0.pt'), map_location='cpu')
    sd1 = torch.load(os.path.join(ck, 'checkpoint-1.pt'), map_location='cpu')
    k = 'image_encoder.trunk.layers.2.blocks.3.mlp.fc1.weight'
    assert k in sd1 and 'text_decoder.trunk.lm_head.weight' in sd1 and sd1[k].shape == (1536, 384)
    assert not torch.equal(sd0[k], sd1[k]) and torch.isfinite(sd1[k]).all()
    mc = get_model_config('cruller_small')
    mc.image_encoder.pretrained = mc.text_decoder.pretrained = False
    model = Cruller(mc, vocab_size=50267)
    model.load_state_dict(sd1)
    assert torch.equal(model.state_dict()[k], sd1[k])
    # a DistributedDataParallel checkpoint carries 'module.' prefixes (ref app/eval.py:135): the task-level loader strips them
    from pixparse_amd.framework import DeviceEnv
    from pixparse_amd.task import TaskCrullerPretrain, TaskCrullerPretrainCfg
    mc2 = get_model_config('cruller_small')
    mc2.image_encoder.pretrained = mc2.text_decoder.pretrained = False
    task = TaskCrullerPretrain(TaskCrullerPretrainCfg(dtype='bfloat16', model=mc2), DeviceEnv())
    task.train_setup(num_batches_per_interval=2)
    state = task.training_state()
    state['model'] = {'module.' + n: v for n, v in sd1.items()}
    task.load_training_state(state)
    assert torch.equal(task.model.state_dict()[k].cpu(), sd1[k])


def test_resume_continues_identically(dev):
    """f-2: training_state() / load_training_state(): an interrupted run continues on the same loss trajectory"""
    from oracle import ref_cpu as R
    from pixparse_amd.framework import DeviceEnv, OptimizationCfg
    from pixparse_amd.task import TaskCrullerPretrain, TaskCrullerPretrainCfg
    _register_test_archs()
    L, layers, img = 24, 2, (37, 50)

    def make():
        cfg = TaskCrullerPretrainCfg(num_intervals=4, num_warmup_intervals=1, eval_frequency=1000, dtype='bfloat16',
                                     opt=OptimizationCfg(learning_rate=1e-3, betas=(0.9, 0.98), clip_grad_value=1.0, clip_grad_mode='norm'),
                                     model=_cfg('vit_test', img, 'RGB', layers, L))
        torch.manual_seed(5)
        t = TaskCrullerPretrain(cfg, DeviceEnv())
        t.train_setup(num_batches_per_interval=2)
        t.train_interval_start()
        return t
    spec = R.ModelSpec('vit_test', 'bart_test', layers, L, img, 3, vocab=50267)
    samples = [R.synthetic_sample(spec, 2, seed=30 + i, ragged=True) for i in range(4)]
    a = make()
    for s in samples[:2]:
        a.train_step(s)
    snap = {k: (v if not isinstance(v, dict) else {kk: (vv.clone() if torch.is_tensor(vv) else vv) for kk, vv in v.items()})
            for k, v in a.training_state().items()}
    snap['model'] = {'module.' + k: v.clone() for k, v in snap['model'].items()}     # as a DDP checkpoint would name them
    la = []
    for s in samples[2:]:
        a.train_step(s)
        la.append(float(a.last_loss))
    b = make()
    b.load_training_state(snap)
    assert b.step == 2 and abs(b.get_current_lr() - R.cosine_lr(2, 1e-3, 2, 8)) < 1e-12
    lb = []
    for s in samples[2:]:
        b.train_step(s)
        lb.append(float(b.last_loss))
    assert all(abs(x - y) < 1e-5 * abs(x) for x, y in zip(la, lb)), (la, lb)


def test_finetune_rvlcdip_steps_vs_oracle_trainer(dev):
    """f-3: cruller_finetune_rvlcdip through TaskFactory -- pretrain checkpoint loaded, 19 tokens added, collate_fn on
    uint8 pages, dict samples, 4-token decoder sequences; loss trajectory and updated weights vs the oracle trainer"""
    import numpy as np
    from oracle import ref_cpu as R
    from pixparse_amd.framework import DeviceEnv, OptimizationCfg
    from pixparse_amd.task import TaskCrullerPretrain, TaskCrullerPretrainCfg, TaskFactory
    _register_test_archs()
    layers, img = 2, (37, 50)
    model_cfg = _cfg('vit_test', img, 'L', layers, 512)
    torch.manual_seed(3)
    pre = TaskCrullerPretrain(TaskCrullerPretrainCfg(dtype='bfloat16', model=model_cfg), DeviceEnv())
    ckpt = {'module.' + k: v.clone() for k, v in pre.model.state_dict().items()}     # a DDP pretrain checkpoint, vocab 50267
    args = dict(num_intervals=2, num_warmup_intervals=0, eval_frequency=1000, dtype='bfloat16', model=model_cfg,
                opt=OptimizationCfg(learning_rate=5e-4, betas=(0.9, 0.98), clip_grad_value=1.0, clip_grad_mode='norm'))
    task, _ = TaskFactory.create_task('cruller_finetune_rvlcdip', args, DeviceEnv(), None)
    task.state_dict = ckpt
    task.resume = True
    torch.manual_seed(4)                                         # the 19 new embedding rows
    task.train_setup(num_batches_per_interval=2)
    V = task.vocab_size
    assert V == 50286
    params = {k: v.detach().cpu().clone() for k, v in task.model.state_dict().items() if not k.endswith('lm_head.weight')}
    for k, v in ckpt.items():                                    # everything but the grown embedding equals the checkpoint
        k = k[7:]
        if k in params and 'embed_tokens' not in k:
            assert torch.equal(params[k], v.cpu()), k
    spec = R.ModelSpec('vit_test', 'bart_test', layers, 512, img, 1, vocab=V)
    tr = R.OracleTrainer(spec, params, lr=5e-4, betas=(0.9, 0.98), eps=1e-6, clip_grad=1.0, accum_steps=1, warmup_t=0, t_initial=4)
    task.train_interval_start()
    rng = np.random.RandomState(0)
    for i in range(3):
        batch = [{'image': rng.randint(0, 256, size=(60 + 7 * j, 45 + 5 * j, 3)).astype(np.uint8), 'label': int(rng.randint(16))} for j in range(3)]
        sample = task.collate_fn(batch)
        assert sample['label'].shape == (3, 4)
        # the oracle shifts by itself: give it the unshifted sequences the collator built
        full = torch.stack([task._tokenize(task._sequence_for(it)) for it in batch])
        full_t = torch.stack([task.text_input_to_target(t) for t in full])
        lo = tr.train_step((sample['image'], full, full_t))
        task.train_step(sample)
        lh = float(task.last_loss)
        assert abs(lh - lo) / abs(lo) < 2e-3, (i, lh, lo)
    k = 'text_decoder.trunk.model.decoder.embed_tokens.weight'
    assert rel(task.model.state_dict()[k][50267:], tr.params[k].detach()[50267:]) < 2e-3     # the new class-token rows learn


@pytest.mark.parametrize('enc,img,fmt,B', [('vit_test', (37, 50), 'RGB', 3), ('swin_test', (64, 64), 'L', 2)])
def test_greedy_generation_with_kv_cache_vs_oracle(dev, enc, img, fmt, B):
    """f-4: utils.ocr_utils.get_generated_tokens (KV cache, skinny projections, single-query attention) against the
    oracle's restatement of the reference loop (whole decoder re-run per token). Random weights give nearly flat
    logits, so besides the per-step logits the check is: every token chosen here is an arg-max of the ORACLE's logits
    for the same prefix up to the bf16 tolerance."""
    from oracle import ref_cpu as R
    from pixparse_amd.framework import DeviceEnv
    from pixparse_amd.models import Cruller
    from pixparse_amd.tokenizers import ByteBartTokenizer
    from pixparse_amd.utils import generate_ocr, get_generated_tokens
    _register_test_archs()
    layers, L, V = 2, 24, 515
    torch.manual_seed(11)
    model = Cruller(_cfg(enc, img, fmt, layers, L), vocab_size=V)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.endswith('.bias'):
                p.normal_(0, 0.05)
            elif p.dim() >= 2:
                p.mul_(4.0)                                   # spread the logits
    params = {k: v.detach().clone() for k, v in model.state_dict().items() if not k.endswith('lm_head.weight')}
    spec = R.ModelSpec(enc, 'bart_test', layers, L, img, 1 if fmt == 'L' else 3, vocab=V)
    image, _, _ = R.synthetic_sample(spec, B, seed=9)

    class Tok:                                                # prompt id 514, eos 2 inside the 515-token test vocabulary
        trunk = ByteBartTokenizer(base_vocab=514)
    Tok.trunk.add_special_tokens({'additional_special_tokens': ['<s_pretrain>']})
    model.to(dev)
    model._ensure_engines()
    model.refresh_shadows(full=True)
    env = DeviceEnv()
    enc_out = model.image_encoder(image.to(dev))
    steps_max = 12
    ids, logits = get_generated_tokens(model, Tok, enc_out, env, steps_max, '<s_pretrain>', return_logits=True)
    assert ids.shape[0] == B and ids[:, 0].tolist() == [514] * B and 1 <= ids.shape[1] <= steps_max + 1
    # oracle: its own encoder output, then the decoder re-run on OUR prefixes (teacher forcing keeps both on one path)
    oenc = R.encode_image(params, spec, image, 'bf16')
    close = lambda a, b, tol: float((a - b).abs().max()) <= tol * max(1.0, float(b.abs().max()))
    assert close(enc_out.float().cpu(), oenc.float(), 3e-2)
    for t in range(len(logits)):
        ol = R.bart_decoder_forward(params, spec.dec_arch, layers, ids[:, :t + 1].cpu(), oenc, 'bf16', prefix='text_decoder.trunk.')[:, -1, :].float()
        mine = logits[t].cpu()
        assert close(mine, ol, 3e-2), (t, float((mine - ol).abs().max()))
        if t + 1 < ids.shape[1]:
            chosen = ids[:, t + 1].cpu()
            margin = ol.max(-1).values - ol.gather(1, chosen[:, None])[:, 0]
            assert float(margin.max()) <= 3e-2 * max(1.0, float(ol.abs().max())), (t, margin)
    # same loop semantics as the oracle's restatement of ocr_utils.py:165-197 when it is fed OUR logits' arg-max path
    oids = R.greedy_generate(params, spec, oenc, 514, 2, steps_max, 'bf16')
    assert oids.shape[1] <= steps_max + 1 and oids[:, 0].tolist() == [514] * B
    # hipGraph replay of the step == eager steps, token for token (same kernels, same arguments)
    g_ids = get_generated_tokens(model, Tok, enc_out, env, steps_max, '<s_pretrain>', use_graph=True)
    e_ids = get_generated_tokens(model, Tok, enc_out, env, steps_max, '<s_pretrain>', use_graph=False)
    assert torch.equal(g_ids, e_ids) and torch.equal(e_ids, ids)
    texts = generate_ocr(model, Tok, enc_out, env, 5, '<s_pretrain>')
    assert len(texts) == B and all(t.startswith('<s_pretrain>') for t in texts)


@pytest.mark.parametrize('accum', [1, 2])
def test_graphed_train_step_equals_eager(dev, accum):
    """TaskCrullerPretrainCfg.graph_step: the micro-step (forward, CE, backward, clip / AdamW / zero_grad) replayed from a hipGraph is the
    eager step bit for bit -- 8 micro-steps with warm-up cosine LR (device-side schedule, crl_optim_prepare), clip-norm and grad
    accumulation 1 / 2 (two graphs: with and without the optimiser tail): identical losses, gradient norms, learning rates taken from
    the device words, and final parameters / AdamW moments"""
    from pixparse_amd.data import synthetic_batch
    from pixparse_amd.framework import DeviceEnv, OptimizationCfg
    from pixparse_amd.task import TaskCrullerPretrain, TaskCrullerPretrainCfg
    _register_test_archs()
    L, layers, img = 24, 2, (37, 50)

    def run(graph):
        cfg = TaskCrullerPretrainCfg(num_intervals=2, num_warmup_intervals=1, eval_frequency=1000, dtype='bfloat16', graph_step=graph,
                                     opt=OptimizationCfg(learning_rate=1e-3, betas=(0.9, 0.98), clip_grad_value=1.0, clip_grad_mode='norm',
                                                         grad_accum_steps=accum),
                                     model=_cfg('vit_test', img, 'RGB', layers, L))
        torch.manual_seed(3)
        task = TaskCrullerPretrain(cfg, DeviceEnv())
        task.train_setup(num_batches_per_interval=4)
        task.train_interval_start()
        assert task._graph_on == graph
        rec = []
        for i in range(8):
            task.train_step(synthetic_batch(2, 3, img, L, task.vocab_size, seed=60 + i, ragged=True))
            st = task.optimizer.state.cpu()
            rec.append((float(task.last_loss), float(st[0]), float(st[8]), float(st[3]), float(st[7]), task.get_current_lr()))
        kinds = {k: type(v).__name__ for k, v in task._graphs.items()}
        ar = task.model.arena
        return rec, ar.p.clone(), ar.m.clone(), ar.v.clone(), kinds
    eager, pe, me, ve, _ = run(False)
    graph, pg, mg, vg, kinds = run(True)
    # graphs are keyed by (optimiser tail, first micro-step of the accumulation window: its weight gradients overwrite the arena)
    assert kinds == ({(True, True): 'CUDAGraph'} if accum == 1 else {(False, True): 'CUDAGraph', (True, False): 'CUDAGraph'})
    assert eager == graph, list(zip(eager, graph))
    assert torch.equal(pe, pg) and torch.equal(me, mg) and torch.equal(ve, vg)
    # the device-side LR is the scheduler's closed form: host mirror (set AFTER the update, for the next one) vs the device word of the update
    n_upd = 8 // accum
    assert eager[-1][3] == n_upd and eager[-1][4] == n_upd
    lrs = [r[2] for r in eager[accum - 1::accum]]
    w, t = 4 // accum, 8 // accum          # one warm-up interval, two intervals in all, 4 // accum updates per interval
    want = [1e-3 * u / w if u < w else 0.5 * 1e-3 * (1 + math.cos(math.pi * u / t)) if u < t else 0.0 for u in range(n_upd)]
    assert all(abs(a - b) <= 1e-9 + 1e-6 * b for a, b in zip(lrs, want)), (lrs, want)


@pytest.mark.parametrize('enc', ['vit_test', 'swin_test'])
@pytest.mark.parametrize('accum', [1, 3])
def test_weight_gradients_overwrite_on_first_micro_step_same_bits(dev, monkeypatch, enc, accum):
    """round 6: on the first micro-step of an accumulation window the weight-gradient GEMMs (and the bias gradients that ride them) overwrite the
    zero-filled gradient arena instead of adding to it (CrullerModel.backward(first_micro=True); PIXPARSE_AMD_WGRAD_OVERWRITE=0 switches it off).
    Same bits: 7 micro-steps with clip-norm at accumulation 1 and 3 (an update mid-interval, a window cut by the interval end), tied embedding
    (LM-head gradient overwrites, the embedding rows are added behind it), ViT and Swin encoders: identical losses, gradient norms, parameters, moments"""
    from pixparse_amd.data import synthetic_batch
    from pixparse_amd.framework import DeviceEnv, OptimizationCfg
    from pixparse_amd.models import cruller as cruller_mod
    from pixparse_amd.task import TaskCrullerPretrain, TaskCrullerPretrainCfg
    _register_test_archs()
    L, layers = 24, 2
    img = (37, 50) if enc == 'vit_test' else (64, 64)

    def run(overwrite):
        monkeypatch.setattr(cruller_mod, '_WGRAD_OVERWRITE', overwrite)
        cfg = TaskCrullerPretrainCfg(num_intervals=2, num_warmup_intervals=1, eval_frequency=1000, dtype='bfloat16', graph_step=False,
                                     opt=OptimizationCfg(learning_rate=1e-3, betas=(0.9, 0.98), clip_grad_value=1.0, clip_grad_mode='norm',
                                                         grad_accum_steps=accum),
                                     model=_cfg(enc, img, 'RGB', layers, L))
        torch.manual_seed(3)
        task = TaskCrullerPretrain(cfg, DeviceEnv())
        task.train_setup(num_batches_per_interval=4)
        rec, seen = [], []
        for i in range(7):
            if i % 4 == 0:
                task.train_interval_start()
            task.train_step(synthetic_batch(2, 3, img, L, task.vocab_size, seed=60 + i, ragged=True))
            st = task.optimizer.state.cpu()
            rec.append((float(task.last_loss), float(st[0]), float(st[3])))
            seen.append(task.model._engines[0].first_micro)
        ar = task.model.arena
        return rec, ar.p.clone(), ar.m.clone(), ar.v.clone(), ar.g.clone(), seen
    off, po, mo, vo, go, seen_off = run(False)
    on, pn, mn, vn, gn, seen_on = run(True)
    assert not any(seen_off)
    assert seen_on == [(i % 4) % accum == 0 for i in range(7)], seen_on
    if enc == 'vit_test':
        assert off == on, list(zip(off, on))
        assert torch.equal(po, pn) and torch.equal(mo, mn) and torch.equal(vo, vn) and torch.equal(go, gn)
    else:       # Swin: the relative-position-bias gradients are summed with float atomics (1e-10 run to run, whatever the switch says)
        assert all(abs(a[0] - b[0]) <= 1e-5 and abs(a[1] - b[1]) <= 1e-5 * max(1.0, abs(a[1])) for a, b in zip(off, on)), list(zip(off, on))
        for x, y in ((po, pn), (mo, mn), (vo, vn), (go, gn)):
            assert torch.allclose(x, y, rtol=1e-4, atol=1e-6)


def test_greedy_generation_with_prompt_prefill_vs_oracle(dev):
    """f-4 (DocVQA / CORD eval loops, ref task/task_cruller_eval_docvqa.py:279-297): generation from a 12-token prompt.  All but the
    prompt's last token go through ONE prefill pass of the decoder that fills the KV cache, then the single-token decode steps take over
    (eager and hipGraph replay).  Against the oracle's restatement of the reference loop, which re-runs the whole decoder on prompt +
    generated tokens for every step: per-step logits, every chosen token an arg-max of the oracle's logits within the bf16 tolerance,
    and the cache rows written by the prefill equal to the rows the token-by-token path writes (same kernels per row up to the GEMM
    shape: bf16 tolerance)."""
    from oracle import ref_cpu as R
    from pixparse_amd.models import Cruller
    _register_test_archs()
    enc, img, fmt, B = 'vit_test', (37, 50), 'RGB', 2
    layers, L, V = 2, 48, 515
    torch.manual_seed(12)
    model = Cruller(_cfg(enc, img, fmt, layers, L), vocab_size=V)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.endswith('.bias'):
                p.normal_(0, 0.05)
            elif p.dim() >= 2:
                p.mul_(4.0)
    params = {k: v.detach().clone() for k, v in model.state_dict().items() if not k.endswith('lm_head.weight')}
    spec = R.ModelSpec(enc, 'bart_test', layers, L, img, 3, vocab=V)
    image, _, _ = R.synthetic_sample(spec, B, seed=10)
    model.to(dev)
    model._ensure_engines()
    model.refresh_shadows(full=True)
    enc_out = model.image_encoder(image.to(dev))
    prompt = [514, 7, 301, 44, 9, 120, 77, 300, 5, 63, 410, 513]          # <s_task> ... <s_answer>: 12 tokens, the same for both samples
    steps_max = 10
    ids, logits = model.generate_greedy(enc_out, prompt, 2, steps_max, use_graph=False, return_logits=True)
    assert ids[:, :12].tolist() == [prompt] * B and 12 <= ids.shape[1] <= 12 + steps_max
    oenc = R.encode_image(params, spec, image, 'bf16')
    close = lambda a, b, tol: float((a - b).abs().max()) <= tol * max(1.0, float(b.abs().max()))
    for t in range(len(logits)):
        ol = R.bart_decoder_forward(params, spec.dec_arch, layers, ids[:, :12 + t].cpu(), oenc, 'bf16', prefix='text_decoder.trunk.')[:, -1, :].float()
        assert close(logits[t].cpu(), ol, 3e-2), (t, float((logits[t].cpu() - ol).abs().max()))
        if 12 + t < ids.shape[1]:
            chosen = ids[:, 12 + t].cpu()
            margin = ol.max(-1).values - ol.gather(1, chosen[:, None])[:, 0]
            assert float(margin.max()) <= 3e-2 * max(1.0, float(ol.abs().max())), (t, margin)
    oids = R.greedy_generate(params, spec, oenc, prompt, 2, steps_max, 'bf16')
    assert oids[:, :12].tolist() == [prompt] * B
    g_ids = model.generate_greedy(enc_out, prompt, 2, steps_max, use_graph=True)
    t_ids = model.generate_greedy(enc_out, torch.tensor([prompt, prompt]), 2, steps_max, use_graph=False)      # [B, P] tensor form
    assert torch.equal(g_ids, ids) and torch.equal(t_ids, ids)
    # the cache after prefill(11 tokens) == the cache after feeding the same 11 tokens one by one
    _, dec, bufs = model._engines
    model.decode_begin(enc_out, 16)
    model.decode_prefill(torch.tensor([prompt[:-1]] * B, device=dev))
    pre = [bufs.t[f'dec.gen.l{i}.kvc'][:, :11].float().clone() for i in range(layers)]
    model.decode_begin(enc_out, 16)
    for t in range(11):
        model.decode_step(torch.full((B, 1), prompt[t], dtype=torch.int64, device=dev))
    for i in range(layers):
        one = bufs.t[f'dec.gen.l{i}.kvc'][:, :11].float()
        assert float((pre[i] - one).abs().max()) <= 3e-2 * max(1.0, float(one.abs().max())), i


def test_eval_ocr_task_step(dev):
    """f-4: cruller_eval_ocr through TaskFactory: checkpoint hand-over, step() = encode + KV-cache generation + CER / WER"""
    from oracle import ref_cpu as R
    from pixparse_amd.framework import DeviceEnv
    from pixparse_amd.models import Cruller
    from pixparse_amd.task import TaskCrullerEvalOCR, TaskFactory
    _register_test_archs()
    layers, L, img = 2, 24, (37, 50)
    model_cfg = _cfg('vit_test', img, 'RGB', layers, L)
    torch.manual_seed(21)
    trained = Cruller(model_cfg, vocab_size=50267)
    ckpt = {'module.' + k: v.clone() for k, v in trained.state_dict().items()}
    task, _ = TaskFactory.create_task('cruller_eval_ocr', dict(model=model_cfg, dtype='bfloat16'), DeviceEnv(), None)
    assert isinstance(task, TaskCrullerEvalOCR) and task.vocab_size == 50267
    task.resume_state_dict = ckpt
    task.setup()
    k = 'text_decoder.trunk.model.decoder.layers.1.fc2.weight'
    assert torch.equal(task.model.state_dict()[k].cpu(), ckpt['module.' + k])
    task.max_recursion_length = 6
    spec = R.ModelSpec('vit_test', 'bart_test', layers, L, img, 3, vocab=50267)
    image, tokens, target = R.synthetic_sample(spec, 3, seed=5, ragged=True)
    tokens = torch.randint(4 + 97, 4 + 122, tokens.shape)                  # printable byte tokens so that the decoded targets are text
    tokens[:, 0] = 50266
    tokens[:, 10:] = 1
    tgt = tokens.clone(); tgt[tgt == 1] = -100; tgt[:, 0] = -100
    out = task.step((image, [[t] for t in tokens], [[t] for t in tgt]))    # loader format: lists of per-page tensors
    m = out['ocr_reconstruction']
    assert m is None or (set(m) <= {'wer', 'cer'} and all(v >= 0 for v in m.values()))
    if m and 'wer' in m:
        avg = task.average_metrics({0: out, 1: out})
        assert avg['ocr_reconstruction'] == {'wer': m['wer'], 'cer': m['cer']}
    assert set(task.state_dict()) == {'model'}
    # framework.evaluate: sweeps the accepted loaders, keeps the task's average
    from pixparse_amd.framework import evaluate

    class _L:
        loader = [(image, [[t] for t in tokens], [[t] for t in tgt])] * 2
    if m and 'wer' in m:
        res = evaluate(task, {'eval': _L, 'train': _L})
        assert set(res) == {'eval'} and res['eval']['average']['ocr_reconstruction']['wer'] == m['wer']


def test_eval_rvlcdip_task_step_gpu(dev):
    """cruller_eval_rvlcdip end to end on the device: 19 class tokens added, 5 KV-cache steps, counts returned"""
    import numpy as np
    from pixparse_amd.framework import DeviceEnv
    from pixparse_amd.task import TaskFactory
    _register_test_archs()
    model_cfg = _cfg('vit_test', (37, 50), 'L', 2, 16)
    torch.manual_seed(8)
    task, _ = TaskFactory.create_task('cruller_eval_rvlcdip', dict(model=model_cfg, dtype='bfloat16'), DeviceEnv(), None)
    task.setup()
    rng = np.random.RandomState(1)
    batch = task.collate_fn([{'image': rng.randint(0, 256, (60, 45)).astype(np.uint8), 'label': int(rng.randint(16))} for _ in range(4)])
    m = task.step(batch)
    c = m['classification']
    assert c['n_valid_samples'] == 4 and 0 <= c['correct_samples'] <= 4


def test_eval_docvqa_and_cord_task_steps_gpu(dev):
    """cruller_eval_docvqa / cruller_eval_cord end to end on the device (random weights): multi-token prompt prefill + KV-cache steps
    inside the reference's string-carried loop.  The generated text is checked against the reference's own way of running it -- the
    UNCACHED decoder re-run on the re-tokenised string for every token (model.text_decoder(**prepare_inputs_for_inference(...)),
    task_cruller_eval_docvqa.py:279-297): every token the cached loop chose is an arg-max of the uncached logits up to the bf16
    tolerance."""
    import numpy as np
    from pixparse_amd.framework import DeviceEnv
    from pixparse_amd.task import TaskFactory
    from pixparse_amd.task.task_cruller_eval_docvqa import generate_string
    _register_test_archs()
    model_cfg = _cfg('vit_test', (37, 50), 'L', 2, 96)
    rng = np.random.RandomState(2)
    torch.manual_seed(9)
    task, _ = TaskFactory.create_task('cruller_eval_docvqa', dict(model=model_cfg, dtype='bfloat16'), DeviceEnv(), None)

    def printable_only(model):
        # the byte-level stand-in tokenizer decodes ids 260..50264 to nothing; with random weights the arg-max would mostly land there and the
        # string would never grow.  Zero those rows of the (tied) output embedding: generation then picks printable ASCII, </s> or tags.
        with torch.no_grad():
            for n, p in model.named_parameters():
                if p.dim() >= 2:
                    p.mul_(4.0)
            E = dict(model.named_parameters())['text_decoder.trunk.model.decoder.embed_tokens.weight']
            E[:2].zero_(); E[3:4 + 32].zero_(); E[4 + 127:50265].zero_()
    printable_only(task.model)
    task.setup()
    items = [{'image': rng.randint(0, 256, (60, 45)).astype(np.uint8), 'labels': {'question': q, 'answers': ['7', 'seven']}, 'image_id': i, 'question_id': i}
             for i, q in enumerate(['how many?', 'total'])]
    batch = task.collate_fn(items)
    task.step(batch)
    assert len(task.all_predictions) == 2 and task.all_ground_truths == [['7', 'seven']] * 2
    anls = task.average_metrics({})['ANLS']
    assert 0.0 <= anls <= 1.0 and task.gen_stats['prefills'] >= 2
    # cached loop vs the reference-style uncached re-forward on one sample, 6 tokens
    tok, model = task.tokenizer.trunk, task.model
    enc = model.image_encoder(batch['images'][:1].to(dev)).clone()
    prompt = '<s_docvqa><s_question>how many?</s_question><s_answer>'
    text = generate_string(model, task.tokenizer, enc, prompt, dev, max_steps=6)
    gen = tok.encode(text, add_special_tokens=False)
    ids = tok.encode(prompt, add_special_tokens=False)
    assert gen[:len(ids)] == ids and len(gen) > len(ids)
    for t in range(len(ids), len(gen)):
        inputs = model.text_decoder.prepare_inputs_for_inference(torch.tensor([gen[:t]], device=dev), enc, tok.pad_token_id)
        ol = model.text_decoder(**inputs)['logits'][0, -1].float()
        margin = float(ol.max() - ol[gen[t]])
        assert margin <= 3e-2 * max(1.0, float(ol.abs().max())), (t, margin)
    # CORD: one-token prompt, nTED / F1 bookkeeping on generated JSON
    task, _ = TaskFactory.create_task('cruller_eval_cord', dict(model=model_cfg, dtype='bfloat16'), DeviceEnv(), None)
    printable_only(task.model)
    task.setup()
    gts = [{'gt_parse': {'menu': [{'nm': 'cake', 'cnt': '2'}], 'total': {'total_price': '9'}}}, {'gt_parse': {'menu': {'nm': 'tea'}}}]
    batch = task.collate_fn([{'image': rng.randint(0, 256, (60, 45)).astype(np.uint8), 'ground_truth': repr(g)} for g in gts])
    m = task.step(batch)
    assert 0.0 <= m['batch_accuracy'] <= 1.0 and len(task.acc_list) == 2
    avg = task.average_metrics({0: m})
    assert set(avg) == {'average_accuracy', 'f1_score'} and 0.0 <= avg['f1_score'] <= 1.0


def test_bench_json_contract(dev, capsys, monkeypatch):
    """bench.py prints ONE JSON line with the driver's keys + roofline (live) on a small config, in-process"""
    import importlib, sys as _sys
    monkeypatch.setattr(_sys, 'argv', ['bench.py', '--model', 'cruller_base_960x640', '--batch', '2', '--steps', '2', '--warmup', '1', '--no-cpu-baseline'])
    bench = importlib.import_module('bench')
    bench.main()
    line = [l for l in capsys.readouterr().out.splitlines() if l.startswith('{')][-1]
    d = json.loads(line)
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data', 'config'):
        assert k in d, k
    assert d['n_gpus'] == 1 and d['steps'] == 2 and d['unit'] == 'docs/s' and d['value'] > 0 and d['scaling'] == 'weak'
    assert 'workload' in d['config'] and 'model' not in d['config']
    r = d['roofline']
    assert r['bound'] == 'mfma' and r['peak'] == 2500.0 and r['kernel'].startswith('attn_') and abs(r['frac'] - r['achieved'] / r['peak']) < 1e-3
    assert r['launches_timed'] > 0 and r['kernels'][r['kernel']]['ms_per_launch'] == r['ms_per_launch']
    assert 500 < r['peak_measured'] < 2500 and abs(r['frac_measured'] - r['achieved'] / r['peak_measured']) < 1e-3
    h = d['host_inputs']                                   # the reference boundary: host batches, H2D inside the step
    assert h['unit'] == 'docs/s' and 0 < h['value'] and d['collectives'] == 'none'
