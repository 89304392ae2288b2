"""HIP path vs the CPU oracle at the REAL widths and depths of BASELINE.json's configs (`-m gpu`).

tests/test_model_gpu.py pins every engine against the oracle at dim 128 / depth 2; here the same comparison runs on the
architectures the headline is quoted on (VERDICT r1 "Next round" item 1):

 (a) full depth, small geometry      ViT-L/14 CLIP (24 blocks, D 1024, 16 heads) + BART-large 10 layers, V = 50267, on a
                                     224x168 image (193 encoder tokens), 128-token targets, batch 2: loss, every parameter
                                     gradient and the total gradient norm against the oracle's bf16 policy, and the drift of
                                     both against the oracle's fp32 policy;
 (b) full length, truncated depth    cfg-3 widths at N = 6189 / T = 1023 (the exact sample bench.py's cpu_baseline times):
                                     2 encoder blocks + 1 decoder layer, batch 1: loss, d(encoder output), weight gradients;
 (c) cfg-2 (cruller_base 960x640, batch 8): determinism, expected initial loss, truncated-depth oracle compare at N = 2401;
 (d) cfg-5 (cruller_large_6layers, 2560x1920 -> N = 24935, T = 2047, micro-batch 2, grad-accum 4): finite, deterministic,
     accumulation-step bookkeeping, attention rows at N = 24935 against fp32 recomputation.

Reference call sites: task/task_cruller_pretrain.py:247-257 (the `_forward` closure), models/cruller.py:14-21.
Tolerances are the ones of tests/test_model_gpu.py: loss 1e-3 relative (BASELINE.json north_star), 5 % relative L2 per
gradient tensor, 2 % on the total gradient norm.
"""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu
BF16, F32 = torch.bfloat16, torch.float32
VOCAB = 50267
VIT_L, BART_L = 'vit_large_patch14_clip_224.datacompxl', 'facebook/bart-large'
VIT_B, BART_B = 'vit_base_patch16_224', 'facebook/bart-base'


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'these tests need the MI355X'
    return torch.device('cuda:0')


def rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def _truncated(enc_name, depth):
    """register `enc_name` cut to `depth` blocks on both sides (product arch table + oracle arch table)"""
    from oracle import ref_cpu as R
    from pixparse_amd.models.archs import VIT_ARCHS, register_arch
    name = f'{enc_name}@depth{depth}'
    arch = dict(VIT_ARCHS[enc_name], depth=depth)
    register_arch('vit', name, arch)
    R.VIT_ARCHS[name] = dict(R.VIT_ARCHS[enc_name], depth=depth)
    return name


def _model_cfg(enc, img, fmt, dec, layers, L):
    from pixparse_amd.models import ImageEncoderCfg, ModelCfg, TextDecoderCfg
    return ModelCfg(image_encoder=ImageEncoderCfg(name=enc, image_fmt=fmt, image_size=img, pretrained=False),
                    text_decoder=TextDecoderCfg(name=dec, pretrained=False, num_decoder_layers=layers, max_length=L))


def _build_pair(dev, enc, img, fmt, dec, layers, L, seed):
    """(HIP model on the device, oracle spec, shared fp32 parameters): both sides start from the oracle's deterministic
    init (N(0, .02) everywhere, biases included, LayerNorm scales 1 + N(0, .02))"""
    from oracle import ref_cpu as R
    from pixparse_amd.models import Cruller
    spec = R.ModelSpec(enc, dec, layers, L, img, 1 if fmt == 'L' else 3, vocab=VOCAB)
    params = R.init_params(spec, seed)
    model = Cruller(_model_cfg(enc, img, fmt, dec, layers, L), vocab_size=VOCAB)
    sd = dict(params)
    sd['text_decoder.trunk.lm_head.weight'] = params['text_decoder.trunk.model.decoder.embed_tokens.weight']
    model.load_state_dict(sd)
    assert {k: tuple(v.shape) for k, v in params.items()} == spec.param_shapes()
    model.to(dev)
    model.arena.alloc_training_state()
    return model, spec, params


def _oracle_grads(spec, params, image, ti, tt, policy, fast_attn=False):
    from oracle import ref_cpu as R
    op = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    loss = R.cruller_loss(op, spec, image, ti, tt, policy, fast_attn=fast_attn)
    loss.backward()
    return float(loss), {k: v.grad for k, v in op.items()}


def _compare_grads(model, ograds, tol_tensor, tol_total):
    worst = []
    for k, og in ograds.items():
        g = model.arena.grad(k)
        if k.endswith('k_proj.bias'):   # identically zero gradient (softmax shift invariance): rounding noise on both sides
            qn = float(ograds[k.replace('k_proj', 'q_proj')].norm())
            assert float(g.norm()) < 5e-2 * qn + 1e-12, k
            continue
        worst.append((rel(g, og), k))
    worst.sort(reverse=True)
    assert worst[0][0] < tol_tensor, worst[:6]
    tot = math.sqrt(sum(float((g.float() ** 2).sum()) for g in ograds.values()))
    mine = float(model.arena.g.norm())
    assert abs(mine - tot) / tot < tol_total, (mine, tot)
    return worst


# ------------------------------------------------------------------------------------------------ (a)
def test_full_depth_real_width_vs_oracle(dev):
    """ViT-L/14 CLIP x 24 + BART-large x 10 at their real widths: bf16 error growth through 24 pre-LN + 10 post-LN layers
    at D = 1024 stays inside the tolerances asserted at toy width"""
    from oracle import ref_cpu as R
    img, L, B = (224, 168), 128, 2
    model, spec, params = _build_pair(dev, VIT_L, img, 'RGB', BART_L, 10, L, seed=11)
    image, tokens, target = R.synthetic_sample(spec, B, seed=5, ragged=True)
    ti, tt = R.shift_tokens(tokens, target)
    with torch.no_grad():
        ologits = R.cruller_forward(params, spec, image, ti, 'bf16')
        loss32 = float(R.cross_entropy(R.cruller_forward(params, spec, image, ti, 'fp32'), tt))
    oloss, ograds = _oracle_grads(spec, params, image, ti, tt, 'bf16')
    out = model(image.to(dev), ti.to(dev))
    assert out['logits'].shape == (B, L - 1, VOCAB)
    assert rel(out['logits'], ologits) < 2e-2
    loss = float(model.forward_loss(image.to(dev), ti.to(dev), tt.to(dev)))
    assert abs(loss - oloss) / oloss < 1e-3, (loss, oloss)
    # drift against exact arithmetic: the HIP path may not be further from fp32 than the reference's own bf16 policy is
    # (plus the 1e-3 the two bf16 executions are allowed to differ by)
    assert abs(loss - loss32) <= abs(oloss - loss32) + 1e-3 * loss32, (loss, oloss, loss32)
    model.backward()
    worst = _compare_grads(model, ograds, 5e-2, 2e-2)
    print(f'\n[a] loss hip {loss:.6f} oracle-bf16 {oloss:.6f} oracle-fp32 {loss32:.6f}; worst grad rel-L2 {worst[0]}')


# ------------------------------------------------------------------------------------------------ (b)
@pytest.mark.parametrize('B', [1, 2])
def test_full_length_truncated_depth_vs_oracle(dev, B):
    """cfg-3 widths at the full sequence lengths (N = 6189 encoder tokens = 97 key tiles with a ragged tail, T = 1023):
    2 encoder blocks + 1 decoder layer + the 50267-column LM head, batch 1 and batch 2 (the batch strides of every kernel at N = 6189
    against the oracle, not only property-checked). The oracle uses torch's fused CPU SDPA for the 6189^2 attention (bench.py's
    cpu_baseline sample), everything else is the parity restatement."""
    from oracle import ref_cpu as R
    enc = _truncated(VIT_L, 2)
    img, L = (1280, 960), 1024
    model, spec, params = _build_pair(dev, enc, img, 'RGB', BART_L, 1, L, seed=12)
    image, tokens, target = R.synthetic_sample(spec, B, seed=6, ragged=True)
    ti, tt = R.shift_tokens(tokens, target)
    # oracle with the encoder output kept, so that d(loss)/d(encoder output) can be compared as well
    op = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    oenc = R.vit_forward(op, spec.enc_arch, image, 'bf16', prefix='image_encoder.trunk.', fast_attn=True)
    oenc.retain_grad()
    ologits = R.bart_decoder_forward(op, spec.dec_arch, 1, ti, oenc, 'bf16', prefix='text_decoder.trunk.', fast_attn=True)
    oloss = R.cross_entropy(ologits, tt)
    oloss.backward()
    loss = float(model.forward_loss(image.to(dev), ti.to(dev), tt.to(dev)))
    assert abs(loss - float(oloss)) / float(oloss) < 1e-3, (loss, float(oloss))
    e, _, bufs = model._engines
    assert e.N == 6189 and bufs.t['vit.norm.y32'].shape == (B * 6189, 1024)
    for b in range(B):
        assert rel(bufs.t['vit.norm.y32'].view(B, 6189, 1024)[b], oenc.detach()[b]) < 1e-2
    model.backward()
    for b in range(B):
        assert rel(bufs.t['denc'].view(B, 6189, 1024)[b], oenc.grad[b]) < 5e-2
    ograds = {k: v.grad for k, v in op.items()}
    worst = _compare_grads(model, ograds, 5e-2, 2e-2)
    print(f'\n[b] loss hip {loss:.6f} oracle {float(oloss):.6f}; worst grad rel-L2 {worst[0]}')


# ------------------------------------------------------------------------------------------------ (b')
def test_cfg3_full_depth_full_length_forward_loss_vs_oracle(dev):
    """BASELINE.json north_star on cfg-3 itself, FORWARD ONLY, BATCH 1: "loss within 1e-3 rel of reference" for cruller_large_1280x960
    with ALL 24 encoder blocks + 10 decoder layers at N = 6189 / T = 1023, V = 50267.  The oracle's bf16-policy forward is ~8.5 TFLOP on
    host cores (torch's fused CPU SDPA for the 6189^2 attention): it was run ONCE (tests/golden/make_g7.py, committed next to its
    output) and this test compares with that fixture -- the loss, 64 rows of the encoder output (what the ten cross-attentions read),
    its norm and column sums, and the logits of six positions; PIXPARSE_AMD_LIVE_ORACLE=1 runs the oracle live instead (minutes on the GPU
    box's host cores) and checks the fixture against it on the way.  The backward at this depth is covered by (a) full depth / short
    sequences and (b) full length / 2 + 1 layers at batch 1 and 2.  ref: task/task_cruller_pretrain.py:247-257."""
    from oracle import ref_cpu as R
    import json
    import os
    from safetensors.torch import load_file
    gdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
    meta = json.load(open(os.path.join(gdir, 'g7_cfg3_forward.json')))
    g7 = load_file(os.path.join(gdir, 'g7_cfg3_forward.safetensors'))
    assert meta['param_seed'] == 14 and meta['sample_seed'] == 8 and meta['N'] == 6189 and meta['T'] == 1023
    img, L, B = (1280, 960), 1024, 1
    model, spec, params = _build_pair(dev, VIT_L, img, 'RGB', BART_L, 10, L, seed=14)
    image, tokens, target = R.synthetic_sample(spec, B, seed=8, ragged=True)
    ti, tt = R.shift_tokens(tokens, target)
    loss = float(model.forward_loss(image.to(dev), ti.to(dev), tt.to(dev)))
    e, d_, bufs = model._engines
    assert e.N == 6189 and d_.T == 1023 and e.a['depth'] == 24 and d_.L == 10
    enc_hip = bufs.t['vit.norm.y32'].float().cpu().clone()
    out = model(image.to(dev), ti.to(dev))                      # forward() leaves clean logits in the buffer (forward_loss overwrote them)
    rows = torch.tensor(meta['logit_rows'])
    logits_hip = out['logits'][0, rows.to(dev)].float().cpu()
    oloss, erows = meta['loss'], torch.tensor(meta['enc_rows'])
    if os.environ.get('PIXPARSE_AMD_LIVE_ORACLE', '0') == '1':
        torch.set_num_threads(os.cpu_count() or 1)
        with torch.no_grad():
            oenc = R.vit_forward(params, spec.enc_arch, image, 'bf16', prefix='image_encoder.trunk.', fast_attn=True)
            ologits = R.bart_decoder_forward(params, spec.dec_arch, 10, ti, oenc, 'bf16', prefix='text_decoder.trunk.', fast_attn=True)
            live = float(R.cross_entropy(ologits, tt))
        assert abs(live - oloss) / oloss < 1e-5 and rel(oenc[0, erows], g7['enc_rows']) < 1e-4 and rel(ologits[0, rows], g7['logit_rows']) < 1e-4
        assert rel(enc_hip, oenc[0]) < 2e-2
    assert abs(loss - oloss) / oloss < 1e-3, (loss, oloss)
    # 24 pre-LN blocks of bf16 GEMMs / attention on 6189 tokens: sampled rows, the norm of the whole output and its column sums
    assert rel(enc_hip[erows], g7['enc_rows']) < 2e-2
    assert abs(float(enc_hip.norm()) - float(g7['enc_norm'])) / float(g7['enc_norm']) < 5e-3
    assert rel(enc_hip.sum(0), g7['enc_colsum']) < 2e-2
    assert rel(logits_hip, g7['logit_rows']) < 3e-2
    print(f'\n[b\'] cfg-3 full depth x full length: loss hip {loss:.6f} oracle {oloss:.6f} (rel {abs(loss - oloss) / oloss:.2e}); '
          f'encoder rows rel-L2 {rel(enc_hip[erows], g7["enc_rows"]):.2e}')


# ------------------------------------------------------------------------------------------------ (b'')
@pytest.mark.parametrize('chain', [0, 13])
def test_cfg3_full_depth_full_length_backward_vs_oracle(dev, chain):
    """The BACKWARD of the north-star configuration itself, batch 1: all 24 encoder blocks + 10 decoder layers at N = 6189 / T = 1023 -- the
    only place where the single-pass attention stream (auto mode selects it for Nq >= 1000) and its key-block chains meet the oracle at full
    depth (VERDICT r4 Weak #1: every link of a chain adds a bf16 rounding to the running dQ; how that compounds through 24 pre-LN blocks and
    10 cross-attentions was measured nowhere).  The oracle side is the fixture G8 (tests/golden/make_g8.py: oracle/ref_cpu.py's autograd
    through the whole model, run once on the GPU box's host cores): loss, total gradient norm, the norm of EVERY parameter gradient and 64
    sampled rows each of four gradients at the far ends of the backward sweep.  chain = 0: the automatic chain length (4 at this shape);
    chain = 13: the longest chain the policy can pick (two workgroups per head).  ref: task/task_cruller_pretrain.py:259-268."""
    from oracle import ref_cpu as R
    from pixparse_amd import hip
    import json
    import os
    from safetensors.torch import load_file
    gdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
    meta = json.load(open(os.path.join(gdir, 'g8_cfg3_backward.json')))
    g8 = load_file(os.path.join(gdir, 'g8_cfg3_backward.safetensors'))
    assert meta['param_seed'] == 14 and meta['sample_seed'] == 8
    img, L, B = (1280, 960), 1024, 1
    model, spec, params = _build_pair(dev, VIT_L, img, 'RGB', BART_L, 10, L, seed=14)
    image, tokens, target = R.synthetic_sample(spec, B, seed=8, ragged=True)
    ti, tt = R.shift_tokens(tokens, target)
    hip.call('crl_attn_bwd_set_chain', chain)
    try:
        loss = float(model.forward_loss(image.to(dev), ti.to(dev), tt.to(dev)))
        model.backward()
        torch.cuda.synchronize()
    finally:
        hip.call('crl_attn_bwd_set_chain', 0)
    assert abs(loss - meta['loss']) / meta['loss'] < 1e-3, (loss, meta['loss'])
    total = float(model.arena.g.norm())
    assert abs(total - meta['total_grad_norm']) / meta['total_grad_norm'] < 2e-2, (total, meta['total_grad_norm'])
    # every parameter gradient: its norm against the oracle's (5 %; the k-projection biases have an identically zero gradient)
    worst_n = []
    for k, on in meta['grad_norms'].items():
        if k.endswith('k_proj.bias'):
            continue
        hn = float(model.arena.grad(k).float().norm())
        worst_n.append((abs(hn - on) / max(on, 1e-30), k))
    worst_n.sort(reverse=True)
    assert worst_n[0][0] < 5e-2, worst_n[:6]
    # sampled rows of the gradients behind the whole backward sweep: relative L2 against the oracle's rows
    worst_r = []
    for name, idx in meta['rows'].items():
        g = model.arena.grad(name).float()
        g2 = g.reshape(-1, g.shape[-1]) if g.dim() != 2 else g
        worst_r.append((rel(g2[torch.tensor(idx, device=g2.device)].cpu(), g8[name]), name))
    worst_r.sort(reverse=True)
    assert worst_r[0][0] < 5e-2, worst_r
    print(f"\n[b''] cfg-3 full depth x full length backward, chain {chain}: loss {loss:.6f} vs {meta['loss']:.6f}; total grad norm {total:.5f} vs "
          f"{meta['total_grad_norm']:.5f}; worst tensor norm {worst_n[0]}; worst sampled rows {worst_r[0]}")


# ------------------------------------------------------------------------------------------------ (a')
def test_cfg1_real_width_swin_tiny_bart_base_vs_oracle(dev):
    """BASELINE.json configs[0] at its REAL widths: swin_tiny_patch4_window7_224 (C = 96 / 192 / 384 / 768, window 7, 3 / 6 / 12 / 24
    heads of 32 channels, depths 2-2-6-2 with shifted windows and three patch mergings) + BART-base (D 768, 12 heads) x 2 layers, 224x224
    images, 128-token targets, batch 2: loss and every parameter gradient against the oracle.  C = 96 is the only user of the
    K % 64 != 0 branch of the 128x128 GEMM kernel (K = 96 -> 32-deep K tiles)."""
    from oracle import ref_cpu as R
    import os
    nthreads = torch.get_num_threads()
    torch.set_num_threads(min(16, os.cpu_count() or 1))     # many small CPU ops (window attention on 49-token windows): 256 threads only add fork / join time
    enc, dec = 'swin_tiny_patch4_window7_224', BART_B
    img, L, B = (224, 224), 128, 2
    model, spec, params = _build_pair(dev, enc, img, 'RGB', dec, 2, L, seed=15)
    image, tokens, target = R.synthetic_sample(spec, B, seed=9, ragged=True)
    ti, tt = R.shift_tokens(tokens, target)
    oloss, ograds = _oracle_grads(spec, params, image, ti, tt, 'bf16')
    with torch.no_grad():
        ologits = R.cruller_forward(params, spec, image, ti, 'bf16')
    out = model(image.to(dev), ti.to(dev))
    assert out['logits'].shape == (B, L - 1, VOCAB)
    assert rel(out['logits'], ologits) < 2e-2
    loss = float(model.forward_loss(image.to(dev), ti.to(dev), tt.to(dev)))
    assert abs(loss - oloss) / oloss < 1e-3, (loss, oloss)
    model.backward()
    worst = _compare_grads(model, ograds, 5e-2, 2e-2)
    torch.set_num_threads(nthreads)
    print(f'\n[a\'] cfg-1 real width: loss hip {loss:.6f} oracle {oloss:.6f}; worst grad rel-L2 {worst[0]}')


# ------------------------------------------------------------------------------------------------ (c)
def _run_task(model_name, batch, steps, accum=1, seed0=100):
    from pixparse_amd.data import synthetic_batch
    from pixparse_amd.framework import DeviceEnv, OptimizationCfg
    from pixparse_amd.models import get_model_config
    from pixparse_amd.task import TaskCrullerPretrain, TaskCrullerPretrainCfg
    mc = get_model_config(model_name)
    mc.image_encoder.pretrained = False
    mc.text_decoder.pretrained = False
    cfg = TaskCrullerPretrainCfg(num_intervals=1, num_warmup_intervals=0, eval_frequency=10 ** 9, dtype='bfloat16',
                                 opt=OptimizationCfg(learning_rate=1e-4, clip_grad_value=1.0, clip_grad_mode='norm', grad_accum_steps=accum),
                                 model=mc)
    torch.manual_seed(0)
    task = TaskCrullerPretrain(cfg, DeviceEnv())
    task.train_setup(num_batches_per_interval=max(steps, 2 * accum))
    task.train_interval_start()
    m = task.model
    out = []
    for i in range(steps):
        sample = synthetic_batch(batch, m.in_chans, m.img_size, m.max_length, task.vocab_size, seed=seed0 + i)
        task.train_step(sample)
        out.append((float(task.last_loss), float(task.optimizer.grad_norm()), task.step))
    shapes = dict(N=m._engines[0].N, T=m._engines[1].T, act_gb=m.activation_bytes() / 2 ** 30)
    del task
    torch.cuda.empty_cache()
    return out, shapes


def test_cfg2_base_batch8_deterministic_and_expected_loss(dev):
    """BASELINE.json configs[1]: cruller_base (ViT-B/16 @ 960x640x1 -> 2401 tokens, BART-base 4L, 512-token targets), batch 8"""
    a, sa = _run_task('cruller_base_960x640', 8, 2)
    b, _ = _run_task('cruller_base_960x640', 8, 2)
    assert a == b, (a, b)
    assert sa['N'] == 2401 and sa['T'] == 511
    expect = math.log(VOCAB) + 0.5 * (0.02 * math.sqrt(768)) ** 2   # tied head on unit-variance LN outputs
    assert abs(a[0][0] - expect) < 0.1 and all(math.isfinite(v) for s in a for v in s[:2]), (a, expect)   # measured 11.03 / 10.98
    assert [s[2] for s in a] == [1, 2]


def test_cfg2_full_length_truncated_depth_vs_oracle(dev):
    """cfg-2 widths (D 768, 12 heads, 1-channel 16x16 patches: K = 256) at N = 2401 / T = 511, 2 + 1 layers, batch 2"""
    from oracle import ref_cpu as R
    enc = _truncated(VIT_B, 2)
    img, L, B = (960, 640), 512, 2
    model, spec, params = _build_pair(dev, enc, img, 'L', BART_B, 1, L, seed=13)
    image, tokens, target = R.synthetic_sample(spec, B, seed=7, ragged=True)
    ti, tt = R.shift_tokens(tokens, target)
    oloss, ograds = _oracle_grads(spec, params, image, ti, tt, 'bf16', fast_attn=True)
    loss = float(model.forward_loss(image.to(dev), ti.to(dev), tt.to(dev)))
    assert abs(loss - oloss) / oloss < 1e-3, (loss, oloss)
    model.backward()
    worst = _compare_grads(model, ograds, 5e-2, 2e-2)
    print(f'\n[c] loss hip {loss:.6f} oracle {oloss:.6f}; worst grad rel-L2 {worst[0]}')


# ------------------------------------------------------------------------------------------------ (d)
def test_cfg5_6layers_2560x1920_accum4(dev):
    """BASELINE.json configs[4] on one GPU: cruller_large_6layers, 2560x1920 -> N = 24935 encoder tokens, 2048-token
    targets, micro-batch 2, grad-accum 4: one optimiser update after four micro-steps, finite and bit-reproducible"""
    a, sa = _run_task('cruller_large_6layers', 2, 4, accum=4)
    assert sa['N'] == 24935 and sa['T'] == 2047, sa
    assert [s[2] for s in a] == [0, 0, 0, 1]                      # `step` advances on the fourth micro-step only
    assert all(math.isfinite(s[0]) for s in a) and math.isfinite(a[-1][1]) and a[-1][1] > 0
    expect = (math.log(VOCAB) + 0.5 * (0.02 * math.sqrt(1024)) ** 2) / 4       # loss / grad_accum_steps (ref :255-256)
    assert all(abs(s[0] - expect) < 0.02 for s in a), (a, expect)
    b, _ = _run_task('cruller_large_6layers', 2, 4, accum=4)
    assert a == b, (a, b)


def test_attention_rows_at_24935_tokens(dev):
    """the cfg-5 encoder attention (390 key tiles, ragged tail of 39 keys): sampled query rows of the forward and of dQ,
    sampled keys of dK / dV against fp32 recomputation; sum_k dV = sum_q dO P^T column identity"""
    from pixparse_amd import ops
    B, H, N, d = 1, 2, 24935, 64
    D, scale = H * d, d ** -0.5
    g = torch.Generator(device=dev).manual_seed(3)
    qkv = torch.randn(B, N, 3 * D, generator=g, device=dev).to(BF16)
    q, k, v = qkv[:, :, :D], qkv[:, :, D:2 * D], qkv[:, :, 2 * D:]
    o = torch.empty(B, N, D, dtype=BF16, device=dev)
    lse = torch.empty(B, H, N, device=dev)
    ops.attn_fwd(q, k, v, o, lse, H, scale, False)
    hd = lambda t: t.float().reshape(B, N, H, d).transpose(1, 2)
    Q, K, V, O = hd(q), hd(k), hd(v), hd(o)
    rows = torch.cat([torch.randperm(N, generator=torch.Generator().manual_seed(0))[:62], torch.tensor([0, N - 1])]).to(dev)
    S = Q[:, :, rows] @ K.transpose(-1, -2) * scale
    P = torch.softmax(S, -1)
    ref_o = P.to(BF16).float() @ V
    assert float((O[:, :, rows] - ref_o).abs().max()) < 2e-2
    assert float((lse[:, :, rows] - torch.logsumexp(S, -1)).abs().max()) < 3e-3
    d_o = torch.randn(B, N, D, generator=g, device=dev).to(BF16)
    dqkv = torch.zeros(B, N, 3 * D, dtype=BF16, device=dev)
    delta = torch.empty(2, B, H, N, device=dev)
    ops.attn_bwd(q, k, v, o, d_o, lse, delta, dqkv[:, :, :D], dqkv[:, :, D:2 * D], dqkv[:, :, 2 * D:], H, scale, False)
    dO = hd(d_o)
    dl = (dO * O).sum(-1)                                                       # [B, H, N]
    dP = dO[:, :, rows] @ V.transpose(-1, -2)
    dS = P * (dP - dl[:, :, rows, None])
    ref_dq = (dS.to(BF16).float() @ K) * scale
    got_dq = hd(dqkv[:, :, :D])[:, :, rows]
    assert rel(got_dq, ref_dq) < 2e-2
    keys = torch.cat([torch.randperm(N, generator=torch.Generator().manual_seed(1))[:62], torch.tensor([0, N - 1])]).to(dev)
    St = (Q @ K[:, :, keys].transpose(-1, -2)) * scale                          # [B, H, N, 64]
    Pt = torch.exp(St - lse[..., None])
    ref_dv = Pt.to(BF16).float().transpose(-1, -2) @ dO
    dPt = dO @ V[:, :, keys].transpose(-1, -2)
    dSt = Pt * (dPt - dl[..., None])
    ref_dk = (dSt.to(BF16).float().transpose(-1, -2) @ Q) * scale
    assert rel(hd(dqkv[:, :, 2 * D:])[:, :, keys], ref_dv) < 2e-2
    assert rel(hd(dqkv[:, :, D:2 * D])[:, :, keys], ref_dk) < 2e-2
