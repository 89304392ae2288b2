#!/usr/bin/env python3
"""bench.py -- docs/sec of the Cruller pretrain step on N MI355X (one process per GPU, RCCL).

    python bench.py --gpus 1 --steps 20 --warmup 5        (the defaults: SURVEY.md 8(d) asks for >= 20 timed steps after >= 5 warm-up)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus 8 --steps K --warmup W

A "step" = one TaskCrullerPretrain.train_step(sample) on a synthetic batch of B docs already resident in HBM
(--host-inputs feeds pinned host batches instead): token shift, forward, CE, backward (+bucketed all-reduce),
unscale/clip/AdamW/zero_grad, LR update.  Workload at N=1 = BASELINE.json configs[2]:
cruller_large (ViT-L/14 CLIP + BART-large 10L) bf16, 1280x960x3, 1024 tokens, batch 8 per GPU.
Prints ONE JSON line (rank 0) with the driver's contract + `roofline` + `cpu_baseline`.

`value` is measured with the synthetic batches already resident in HBM; the reference's boundary hands over HOST tensors
(task/task_cruller_pretrain.py:237-242), so a second, shorter timed region feeds pinned host batches (one 118 MB image
batch per step over PCIe inside the step) and is reported next to it as `host_inputs` -- never as `value`.

`roofline` is measured LIVE: libcruller_hip brackets every launch of the attention kernels inside the K timed steps with
HIP events on the launch stream (crl_prof_begin / crl_prof_end); the kernel symbol with the largest share of the step is
reported (achieved = sum of algorithmic FLOPs / sum of launch durations, peak = 2.5 PFLOP/s dense bf16). These are the
launches `rocprofv3 --kernel-trace --stats` averages (profiles/). `roofline.standalone` keeps back-to-back timings of
single kernels on the live buffers (incl. the fc1 GEMM).  `roofline.peak_measured` = the sustained rate of this
library's own 8192^3 bf16 GEMM run back to back inside bench.py (the chip holds ~1.6-1.7 GHz under MFMA load, not the
2.4 GHz behind the datasheet number); `roofline.traffic` = HBM bytes per launch of the dominant kernel from the committed
rocprofv3 --pmc passes (profiles/r2_pmc_traffic.json: 2 x FETCH_SIZE + WRITE_SIZE, the gfx950 correction of
MI355X_MICROARCH.md "HBM").  `cpu_baseline` = oracle/ref_cpu.py on the host cores, bounded
sample, N=1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

PEAK_BF16_TFLOPS = 2500.0   # MI355X dense bf16 MFMA (MI355X_MICROARCH.md "Chip-level parameters")
METRIC = 'docs/sec (whole node) cruller_large bf16 1280x960, 1/2/4/8 MI355X'


def flops_per_doc(task):
    """algorithmic forward FLOPs per document (SURVEY §8d convention: 2mnk per GEMM, attention 4*Nq*Nk*D per
    layer, causal self-attention at 1/2, elementwise ignored); train = 3x."""
    m = task.model
    V = m.vocab_size
    T = m.max_length - 1
    da = m.dec_arch
    D, F, L = da['d_model'], da['ffn'], m.n_layers
    if m.enc_kind == 'vit':
        a = m.enc_arch
        P, De, dep = a['patch'], a['dim'], a['depth']
        gh, gw = m.img_size[0] // P, m.img_size[1] // P
        N = gh * gw + 1
        enc = dep * (24 * N * De * De + 4 * N * N * De) + 2 * (N - 1) * (P * P * m.in_chans) * De
        S = N
    else:
        from pixparse_amd.layers.engines import SwinEngine
        a = m.enc_arch
        enc = 0
        geo = SwinEngine.stage_geometry(a, m.img_size)
        for si, (Hf, Wf, C, heads, w, depth) in enumerate(geo):
            n = Hf * Wf
            enc += depth * (24 * n * C * C + 4 * n * (w * w) * C)
            if si > 0:
                enc += 2 * n * (2 * C) * C
        enc += 2 * geo[0][0] * geo[0][1] * (a['patch'] ** 2 * m.in_chans) * a['embed_dim']
        S = geo[-1][0] * geo[-1][1]
    dec = L * (8 * T * D * D + 4 * T * D * D + 4 * S * D * D + 4 * T * D * F + 2 * T * T * D + 4 * T * S * D) + 2 * T * D * V
    return float(enc + dec), S


def time_kernel(fn, iters=3):
    """average duration (ms) of `fn` (one or more launches on the current stream) by HIP events"""
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


# (the non-causal forward of a prescaled q is the hand-placed stream attn_fwd4w_kernel<2> for Nk >= 128, attn_fwd_pre_kernel<false> below that)
KERNEL_NAMES = ['attn_fwd4w_kernel<2>', 'attn_fwd_pre_kernel<true>', 'attn_bwd_dkdv_kernel<false>', 'attn_bwd_dkdv_kernel<true>',
                'attn_bwd_dq_kernel<false>', 'attn_bwd_dq_kernel<true>', 'attn_bwd_spx_kernel', 'attn_dq_reduce_kernel']


def collect_live_profile(steps):
    """roofline object from the in-library HIP-event timing of the TIMED steps: per kernel symbol the launches, their
    summed duration and their summed algorithmic FLOPs; the dominant one (largest share of the step) is reported --
    these are the same launches `rocprofv3 --kernel-trace --stats` averages in profiles/."""
    import ctypes
    from pixparse_amd import hip
    n = len(KERNEL_NAMES)
    launches = (ctypes.c_int * n)()
    ms = (ctypes.c_double * n)()
    work = (ctypes.c_double * n)()
    hip.call('crl_prof_end', n, ctypes.addressof(launches), ctypes.addressof(ms), ctypes.addressof(work))
    table = {}
    for i, name in enumerate(KERNEL_NAMES):
        if launches[i]:
            tf = work[i] / (ms[i] * 1e-3) / 1e12
            table[name] = {'launches_per_step': round(launches[i] / steps, 1), 'ms_per_launch': round(ms[i] / launches[i], 4),
                           'ms_per_step': round(ms[i] / steps, 2), 'tflops': round(tf, 1), 'frac': round(tf / PEAK_BF16_TFLOPS, 4)}
    if not table:
        return None
    i = max(range(n), key=lambda j: ms[j])
    achieved = work[i] / (ms[i] * 1e-3) / 1e12
    # the WHOLE attention backward (every backward launch of the step, both passes or the fused pass): algorithmic FLOPs = 2 x forward
    # (dV, dP, dK, dQ; recomputed products not credited) over the summed duration
    bwd = [j for j, nm in enumerate(KERNEL_NAMES) if ('bwd' in nm or 'dq_reduce' in nm) and launches[j]]     # the slab reduce of the single-pass form is part of the backward
    bwd_ms, bwd_work = sum(ms[j] for j in bwd), sum(work[j] for j in bwd)
    fwd = [j for j, nm in enumerate(KERNEL_NAMES) if 'fwd' in nm and launches[j]]
    att = {'attention_bwd_total': {'ms_per_step': round(bwd_ms / steps, 2), 'tflops': round(bwd_work / (bwd_ms * 1e-3) / 1e12, 1),
                                   'frac': round(bwd_work / (bwd_ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4),
                                   'what': 'all attention-backward launches of the step; algorithmic FLOPs = 8 Nq Nk 64 per head (2 x forward)'}} if bwd_ms else {}
    if fwd:
        fms, fw = sum(ms[j] for j in fwd), sum(work[j] for j in fwd)
        att['attention_fwd_total'] = {'ms_per_step': round(fms / steps, 2), 'tflops': round(fw / (fms * 1e-3) / 1e12, 1),
                                      'frac': round(fw / (fms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4)}
    return {'bound': 'mfma', 'kernel': KERNEL_NAMES[i], 'achieved': round(achieved, 1), 'peak': PEAK_BF16_TFLOPS, 'unit': 'TFLOP/s',
            'frac': round(achieved / PEAK_BF16_TFLOPS, 4), 'traffic': None, 'ms_per_launch': round(ms[i] / launches[i], 4),
            'launches_timed': int(launches[i]), 'measured': 'HIP events around every launch inside the timed steps (crl_prof_begin/end)',
            'kernels': table, **att}


def measured_gemm_peak(dev, launches=200, reps=2):
    """sustained TFLOP/s of the library's 8192^3 bf16 NT GEMM, back to back (power-limited steady state): the
    microbenchmark peak SURVEY §8(d) asks to be reported next to the 2.5 PFLOP/s datasheet number"""
    from pixparse_amd import ops
    n = 8192
    x = torch.randn(n, n, device=dev).to(torch.bfloat16)
    w = torch.randn(n, n, device=dev).to(torch.bfloat16)
    out = torch.empty(n, n, dtype=torch.bfloat16, device=dev)
    best = 0.0
    for _ in range(reps):
        ms = time_kernel(lambda: ops.linear_fwd(x, w, None, out), iters=launches)
        best = max(best, 2.0 * n ** 3 / (ms * 1e-3) / 1e12)
    return best


KERNEL_SOURCE = 'attention.hip'     # where every kernel of KERNEL_NAMES lives


def pmc_traffic(kernel, grid_threads=None):
    """HBM bytes per launch of `kernel` from the committed PMC passes (scripts/pmc_traffic.py -> profiles/), for the launch geometry
    `grid_threads` when the file splits the symbol by shape.  None when the kernel's source has changed since the passes were taken
    (the file records the sha1 of every kernel source): a stale figure is worse than none."""
    import glob
    import hashlib
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc_traffic.json')))
    if not files:
        return None, None
    with open(files[-1]) as f:
        t = json.load(f)
    src = os.path.join(ROOT, 'pixparse_amd', 'csrc', KERNEL_SOURCE)
    want = t.get('csrc_sha1', {}).get(KERNEL_SOURCE)
    if want is None or not os.path.exists(src) or hashlib.sha1(open(src, 'rb').read()).hexdigest() != want:
        return None, f'{os.path.basename(files[-1])} was measured on an older {KERNEL_SOURCE}: omitted'
    k = t.get('kernels', {}).get(kernel)
    if not k:
        return None, None
    if grid_threads is not None and str(grid_threads) in k.get('by_grid', {}):
        k = k['by_grid'][str(grid_threads)]
    return k['hbm_bytes_per_launch'], t.get('source')


def dominant_kernel_roofline(task, B):
    """time the three heaviest kernels of the step standalone on the live activation buffers of encoder block 0 and
    report the one with the largest share of the step (launch count x duration)."""
    from pixparse_amd import ops
    m = task.model
    enc, dec, bufs = m._engines
    if m.enc_kind != 'vit':
        return None
    T = bufs.t
    D, H, N, F = enc.D, enc.heads, enc.N, enc.F
    depth = enc.a['depth']
    M = B * N
    scale = (D // H) ** -0.5
    qkv, o, lse = T['vit.b0.qkv'], T['vit.b0.o'], T['vit.b0.lse']
    q3 = qkv.view(B, N, 3 * D)
    do = torch.randn(B, N, D, device=qkv.device).to(torch.bfloat16)
    dqkv = torch.empty_like(qkv).view(B, N, 3 * D)
    delta = torch.empty(2, B, H, N, device=qkv.device)
    o2 = torch.empty_like(o)
    lse2 = torch.empty_like(lse)
    h2, act, pre = T['vit.b0.ln2.y16'], torch.empty_like(T['vit.b0.act']), torch.empty_like(T['vit.b0.dact'])
    w1, b1 = enc.W('blocks.0.mlp.fc1.weight'), enc.P('blocks.0.mlp.fc1.bias')
    cand = {}
    t = time_kernel(lambda: ops.attn_fwd(q3[:, :, :D], q3[:, :, D:2 * D], q3[:, :, 2 * D:], o2.view(B, N, D), lse2, H, scale, False, q_prescaled=True))
    cand['attn_fwd4w_kernel<2> (ViT MHSA fwd)'] = (t, 4.0 * N * N * D * B, depth)
    from pixparse_amd import hip
    bwd = lambda: ops.attn_bwd(q3[:, :, :D], q3[:, :, D:2 * D], q3[:, :, 2 * D:], o.view(B, N, D), do, lse, delta,
                               dqkv[:, :, :D], dqkv[:, :, D:2 * D], dqkv[:, :, 2 * D:], H, scale, False, q_prescaled=True)
    bwd()   # fills the delta / lse row constants once
    # algorithmic backward = 4 products (dV, dP, dK, dQ = 2x forward); the dK/dV pass carries 3 of them (+ S recomputed),
    # the dQ pass 1 (+ S, dP recomputed): executed 4 and 3 products.
    hip.call('crl_attn_bwd_set_parts', 2)
    t = time_kernel(bwd)
    cand['attn_bwd_dkdv_kernel<false> (ViT MHSA bwd, dK/dV pass)'] = (t, 6.0 * N * N * D * B, depth)
    hip.call('crl_attn_bwd_set_parts', 4)
    t = time_kernel(bwd)
    cand['attn_bwd_dq_kernel<false> (ViT MHSA bwd, dQ pass)'] = (t, 2.0 * N * N * D * B, depth)
    hip.call('crl_attn_bwd_set_parts', 7)
    t = time_kernel(lambda: ops.linear_fwd(h2, w1, b1, act, ops.EPI_BF16_GELU, aux=pre))
    # per block: qkv (3D) + proj (D) + fc1 (F) + fc2 (F) columns forward, twice that again in backward
    gemm_equiv = 3.0 * (3 * D + D + 2 * F) / F
    cand['gemm256_kernel<NT,GELU> (fc1 49512x4096x1024; stands for all encoder GEMM launches)'] = (t, 2.0 * M * F * D, depth * gemm_equiv)
    best = max(cand.items(), key=lambda kv: kv[1][0] * kv[1][2])
    name, (ms, flops, cnt) = best
    achieved = flops / (ms * 1e-3) / 1e12
    table = {k: {'ms': round(v[0], 3), 'tflops': round(v[1] / (v[0] * 1e-3) / 1e12, 1), 'launch_equiv_per_step': round(v[2], 1)} for k, v in cand.items()}
    return {'bound': 'mfma', 'kernel': name, 'achieved': round(achieved, 1), 'peak': PEAK_BF16_TFLOPS, 'unit': 'TFLOP/s',
            'frac': round(achieved / PEAK_BF16_TFLOPS, 4), 'traffic': None, 'ms_per_launch': round(ms, 3), 'candidates': table}


def cpu_baseline(model_name, flops_train_per_doc):
    """oracle (CPU restatement, bf16 policy, torch SDPA for attention) timed on the host cores on a bounded sample:
    the same architecture at full sequence lengths, batch 1, truncated to 2 encoder blocks + 1 decoder layer,
    one forward+backward; docs/s extrapolated by algorithmic FLOPs."""
    from oracle import ref_cpu as R
    from pixparse_amd.models import get_model_config
    cfg = get_model_config(model_name)
    ie, td = cfg.image_encoder, cfg.text_decoder
    if ie.name not in R.VIT_ARCHS:
        return None
    arch = dict(R.VIT_ARCHS[ie.name], depth=2)
    R.VIT_ARCHS['_bench_trunc'] = arch
    spec = R.ModelSpec('_bench_trunc', td.name, 1, td.max_length, tuple(ie.image_size), 1 if ie.image_fmt == 'L' else 3, vocab=50267)
    cores = os.cpu_count() or 1
    torch.set_num_threads(cores)
    params = {k: v.requires_grad_(True) for k, v in R.init_params(spec, 0).items()}
    image, tokens, target = R.synthetic_sample(spec, 1)
    ti, tt = R.shift_tokens(tokens, target)
    P, De = arch['patch'], arch['dim']
    N = (spec.img_size[0] // P) * (spec.img_size[1] // P) + 1
    T, D, F, V = spec.max_length - 1, spec.dec_arch['d_model'], spec.dec_arch['ffn'], spec.vocab
    f_fwd = 2 * (24 * N * De * De + 4 * N * N * De) + 2 * (N - 1) * (P * P * spec.in_chans) * De + \
        (12 * T * D * D + 4 * N * D * D + 4 * T * D * F + 2 * T * T * D + 4 * T * N * D) + 2 * T * D * V
    t0 = time.time()
    loss = R.cruller_loss(params, spec, image, ti, tt, 'bf16', fast_attn=True)
    loss.backward()
    dt = time.time() - t0
    docs_per_s = (3 * f_fwd / dt) / flops_train_per_doc
    return {'value': round(docs_per_s, 5), 'unit': 'docs/s', 'cores': cores, 'kind': 'port',
            'sample': f'oracle/ref_cpu.py bf16 policy, {model_name} widths at full lengths (N={N}, T={T}), batch 1, truncated to 2 encoder '
                      f'blocks + 1 decoder layer + LM head, one fwd+bwd = {3 * f_fwd / 1e12:.2f} TFLOP in {dt:.1f} s, '
                      f'extrapolated by algorithmic FLOPs to the full {flops_train_per_doc / 1e12:.2f} TFLOP/doc'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--model', default='cruller_large_1280x960')
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--host-inputs', action='store_true', help='feed pinned host batches (PCIe-inclusive rate) instead of HBM-resident ones')
    ap.add_argument('--no-host-leg', action='store_true', help='skip the second timed region with pinned host batches')
    ap.add_argument('--no-peak', action='store_true', help='skip the standalone-kernel and 8192^3 GEMM legs (clean rocprofv3 traces of the step)')
    ap.add_argument('--occupy-cus', type=int, default=0, help='measurement aid: N CUs held by a sleeping side-stream kernel during the timed '
                    'steps (single-GPU stand-in for the CUs RCCL takes in a data-parallel run); the line says so in `disturbance`')
    ap.add_argument('--gemm-schedule', choices=['dynamic', 'static'], default='dynamic', help='tile schedule of the persistent GEMMs (A/B)')
    ap.add_argument('--graph-step', choices=['auto', 'on', 'off'], default='auto', help='replay the micro-step from a hipGraph (auto: models under 250 M parameters)')
    ap.add_argument('--reserved-cus', type=int, default=0, help='persistent GEMMs launch on 256 - N CUs (A/B with --occupy-cus)')
    args = ap.parse_args()

    from pixparse_amd.data import SyntheticLoaderBundle
    from pixparse_amd.framework import DeviceEnv, OptimizationCfg, random_seed
    from pixparse_amd.task import TaskCrullerPretrain, TaskCrullerPretrainCfg
    import torch.distributed as dist

    env = DeviceEnv()
    assert env.device.type == 'cuda', 'bench.py needs MI355X devices'
    assert env.world_size == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={env.world_size}'
    random_seed(42, 0)   # same initial weights on every rank (rank-0 broadcast follows anyway)
    from pixparse_amd.tokenizers import BYTE_TOKENIZER, TokenizerCfg
    cfg = TaskCrullerPretrainCfg(model_name=args.model, dtype='bfloat16', num_intervals=30, num_warmup_intervals=1, eval_frequency=10 ** 9,
                                 opt=OptimizationCfg(learning_rate=3e-4, betas=(0.9, 0.98), clip_grad_value=1.0, clip_grad_mode='norm'),
                                 tokenizer=TokenizerCfg(name=BYTE_TOKENIZER),   # synthetic token ids: only vocab size + special ids matter
                                 # the live roofline brackets kernel launches with HIP events, which cannot be read back from inside a
                                 # graph replay: 'auto' keeps eager launches whenever the roofline leg is on
                                 graph_step={'auto': None if args.no_roofline else False, 'on': True, 'off': False}[args.graph_step])
    cfg.model.image_encoder.pretrained = False    # random-init weights of the named architecture (no checkpoints offline): stated in `data`
    cfg.model.text_decoder.pretrained = False
    task = TaskCrullerPretrain(cfg, env)
    m = task.model
    nb = args.steps + args.warmup
    loader = SyntheticLoaderBundle(batch_size=args.batch, num_batches=nb, in_chans=m.in_chans, img_size=m.img_size,
                                   max_length=m.max_length, vocab_size=task.vocab_size, seed=42, rank=env.global_rank,
                                   device=None if args.host_inputs else env.device)
    task.train_setup(num_batches_per_interval=max(nb, 100))
    task.train_interval_start()
    it = iter(loader.loader)

    def sync():
        torch.cuda.synchronize()
        if env.distributed:
            dist.barrier()
            torch.cuda.synchronize()

    from pixparse_amd import hip, ops
    import contextlib
    ops.gemm_set_schedule(args.gemm_schedule == 'dynamic')
    ops.gemm_set_reserved_cus(args.reserved_cus)
    for _ in range(args.warmup):
        task.train_step(next(it))
    sync()
    task.reducer.reset_stats()
    disturb = ops.OccupyCUs(args.occupy_cus, max_seconds=100.0) if args.occupy_cus > 0 else contextlib.nullcontext()
    disturb.__enter__()
    live = env.global_rank == 0 and not args.no_roofline
    if live:   # HIP events around every attention-kernel launch of the timed steps, on the launch stream (include/crl.h)
        hip.call('crl_prof_begin', 256 * args.steps)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        task.train_step(next(it))
    if args.occupy_cus > 0:     # a device-wide synchronize would wait for the sleepers themselves: drain the compute stream, then release them
        torch.cuda.current_stream().synchronize()
        dt_disturbed = time.perf_counter() - t0
        disturb.__exit__(None, None, None)
    sync()
    dt = dt_disturbed if args.occupy_cus > 0 else time.perf_counter() - t0
    loss = float(task.last_loss)      # the loss of the timed region (the host-input leg below runs further steps)
    live_prof = collect_live_profile(args.steps) if live else None
    comm = task.reducer.stats() if env.distributed else None      # events of the timed steps (the device is synchronised)
    rank_ms = None
    if env.distributed:
        # per-rank step time and exposed communication: min / max over the ranks tell a straggler from a uniformly slow collective
        mine = torch.tensor([dt, -dt, comm['comm_exposed_ms'], -comm['comm_exposed_ms']], device=env.device, dtype=torch.float64)
        dist.all_reduce(mine, op=dist.ReduceOp.MAX)
        rank_ms = {'step_ms_rank_min': round(-float(mine[1]) / args.steps * 1e3, 2), 'step_ms_rank_max': round(float(mine[0]) / args.steps * 1e3, 2),
                   'comm_exposed_ms_rank_min': round(-float(mine[3]), 3), 'comm_exposed_ms_rank_max': round(float(mine[2]), 3)}
        dt = float(mine[0])
    host = None
    if not args.host_inputs and not args.no_host_leg:
        # the reference boundary: `sample` arrives as host tensors and train_step moves it (ref :237-242)
        hs = max(1, min(args.steps, 5))
        hl = SyntheticLoaderBundle(batch_size=args.batch, num_batches=hs + 1, in_chans=m.in_chans, img_size=m.img_size,
                                   max_length=m.max_length, vocab_size=task.vocab_size, seed=43, rank=env.global_rank, device=None)
        hit = iter(hl.loader)
        task.train_step(next(hit))
        sync()
        h0 = time.perf_counter()
        for _ in range(hs):
            task.train_step(next(hit))
        sync()
        hdt = time.perf_counter() - h0
        if env.distributed:
            tt = torch.tensor([hdt], device=env.device, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            hdt = float(tt)
        host = {'value': round(hs * args.batch * env.world_size / hdt, 4), 'unit': 'docs/s', 'ms_per_step': round(hdt / hs * 1e3, 2), 'steps': hs,
                'what': 'same step fed pinned HOST batches (H2D of the image batch inside train_step, the reference boundary); not `value`'}
    docs = args.steps * args.batch * env.world_size
    value = docs / dt
    f_fwd, S = flops_per_doc(task)
    f_train = 3.0 * f_fwd
    step_tflops = value * f_train / 1e12 / env.world_size

    out = {
        'metric': METRIC, 'value': round(value, 4), 'unit': 'docs/s', 'n_gpus': env.world_size, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': round(dt / args.steps * 1e3, 2), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': 'bf16', 'data': 'synthetic (N(0,1) images, uniform random full-length targets), random-init weights, inputs ' +
        ('in pinned host memory (H2D inside the step)' if args.host_inputs else 'resident in HBM'),
        'config': {'workload': f'{args.model}: {m.cfg.image_encoder.name} @ {m.img_size[0]}x{m.img_size[1]}x{m.in_chans} + '
                               f'{m.cfg.text_decoder.name} {m.n_layers}L, {m.max_length}-token targets, V={task.vocab_size}, '
                               f'per-GPU batch {args.batch}, AdamW + clip-norm 1.0, dropout off',
                   'global_batch': args.batch * env.world_size, 'parallelism': f'dp{env.world_size}',
                   'train_tflop_per_doc': round(f_train / 1e12, 3)},
        'loss': round(loss, 5),
        'non_attention_ms_per_step': None,
        'gemm_round_model_us': dict(zip(('a', 'b_per_1024_k', 'calibrated'), getattr(task, 'gemm_model', (None, None, False)))),
        'step_mfma_frac': round(step_tflops / PEAK_BF16_TFLOPS, 4), 'step_tflops_per_gpu': round(step_tflops, 1),
        'activation_gb': round(m.activation_bytes() / 2 ** 30, 2),
        'collectives': ({'nccl': 'rccl'}.get(dist.get_backend(), dist.get_backend()) if env.distributed else 'none'),
        'launch': 'hipGraph replay of the micro-step' if getattr(task, '_graph_on', False) else 'eager launches',
    }
    if host is not None:
        out['host_inputs'] = host
    if comm is not None:
        out['comm'] = dict(comm, **(rank_ms or {}), what='bucketed all-reduce of the gradient arena (torch.distributed nccl = RCCL); comm_exposed_ms = per optimiser '
                           'step, time the compute stream waits in reducer.finish() for collectives still running when backward is done (rank 0: mean and max over its steps; *_rank_min / *_rank_max: over the ranks)')
    if args.occupy_cus or args.reserved_cus or args.gemm_schedule != 'dynamic':
        out['disturbance'] = {'occupied_cus': args.occupy_cus, 'reserved_cus': args.reserved_cus, 'gemm_schedule': args.gemm_schedule,
                              'what': 'A/B run for the multi-GPU CU-contention experiment (DESIGN.md (e)); not a headline number'}
    if env.global_rank == 0:
        if live_prof is not None:
            out['roofline'] = live_prof
            if m.enc_kind == 'vit':     # key blocks per workgroup of the single-pass attention backward (DESIGN.md "Chains of key blocks")
                from pixparse_amd import hip as _hip
                enc0 = m._engines[0]
                live_prof['attn_bwd_chain'] = {'key_blocks': (enc0.N + 255) // 256, 'chain': _hip.query('crl_attn_bwd_chain_for', enc0.N, args.batch * enc0.heads),
                                            'remainder_split_by_query_halves': bool(_hip.query('crl_attn_bwd_qsplit_for', enc0.N, args.batch * enc0.heads))}
            # everything that is not an attention launch (GEMMs, LayerNorm, loss, optimiser, gaps): the line VERDICT r3 asked to watch
            att_ms = sum(k['ms_per_step'] for k in live_prof['kernels'].values())
            out['non_attention_ms_per_step'] = round(out['ms_per_step'] - att_ms, 2)
            # the dominant symbol also serves the decoder's cross-attention; traffic is quoted for the encoder-shape launches only
            # (shape key of scripts/pmc_traffic.py: grid = query tiles x B x H workgroups of 256 threads -- for the dK/dV pass the grid of
            # the dQ pass launched right before it) next to that shape's algorithmic bytes (each operand once)
            grid, algo = None, None
            if m.enc_kind == 'vit' and live_prof['kernel'] in ('attn_fwd4w_kernel<2>', 'attn_bwd_dkdv_kernel<false>', 'attn_bwd_dq_kernel<false>', 'attn_bwd_spx_kernel'):
                enc_ = m._engines[0]
                spx = live_prof['kernel'] == 'attn_bwd_spx_kernel'      # workgroups of 256 keys instead of 128-row tiles
                tiles = (enc_.N + 255) // 256 if (spx or live_prof['kernel'] == 'attn_fwd4w_kernel<2>') else (enc_.N + 127) // 128
                grid = tiles * args.batch * enc_.heads * 256
                # each operand once: q, k, v, dO in, dK, dV out -- and for the single pass dQ, whose per-key-block bf16 partials (ceil(N / 256)
                # slabs the size of dQ, summed by attn_dq_reduce_kernel) are the price of one recomputation instead of two: they are NOT algorithmic
                n_operands = {'attn_fwd4w_kernel<2>': 4, 'attn_bwd_dkdv_kernel<false>': 6, 'attn_bwd_dq_kernel<false>': 6, 'attn_bwd_spx_kernel': 7}[live_prof['kernel']]
                algo = n_operands * args.batch * enc_.N * enc_.D * 2
                shape_note = f'encoder self-attention launches (N = {enc_.N}, grid {grid} threads)'
                if spx:
                    # the decoder's cross-attention launches have the SAME grid (ceil(Nk / 256) key blocks x B x H): the PMC mean is over the
                    # mix of encoder and cross launches, so the algorithmic figure is the mean over the same mix
                    dec_ = m._engines[1]
                    cross = (3 * dec_.T + 4 * enc_.N) * args.batch * enc_.D * 2        # q, dO, dQ rows of the targets; k, v, dK, dV rows of the encoder
                    ne, nd = enc_.a['depth'], dec_.L
                    algo = (ne * algo + nd * cross) // (ne + nd)
                    grid = None                                                      # one grid (chains of key blocks x B x H) for both: the symbol's mean
                    shape_note = (f'mean over the {ne} encoder (N = {enc_.N}) and {nd} cross-attention ({dec_.T} x {enc_.N}) launches of a step, which share one grid; '
                                  f'the bytes above the algorithmic ones are the running partial-dQ slabs: every 256-key block writes its tile rows once and reads '
                                  f'those of the block before it in its chain (by design: DESIGN.md (d))')
            traffic, src = pmc_traffic(live_prof['kernel'], grid)
            if traffic is not None:
                live_prof['traffic'] = traffic
                live_prof['traffic_source'] = src
                if algo:
                    live_prof['traffic_algorithmic'] = algo
                    live_prof['traffic_shape'] = shape_note
            elif src:
                live_prof['traffic_note'] = src
            if env.world_size == 1 and not args.no_peak:
                try:
                    out['roofline']['standalone'] = dominant_kernel_roofline(task, args.batch)
                except Exception as e:  # never lose the headline number to the microbench
                    out['roofline']['standalone'] = {'error': repr(e)}
                try:
                    pk = measured_gemm_peak(env.device)
                    live_prof['peak_measured'] = round(pk, 1)
                    live_prof['frac_measured'] = round(live_prof['achieved'] / pk, 4)
                    live_prof['peak_measured_what'] = ('this library\'s 8192^3 bf16 GEMM (round 5: the 4-wave one-wave-per-SIMD kernel, grouped tile order, overlapped epilogue), '
                                                       '200 launches back to back (sustained clock); the vendor GEMM on the same boxes: profiles/r5_yardstick.txt')
                    out['step_frac_of_measured_peak'] = round(step_tflops / pk, 4)
                except Exception as e:
                    live_prof['peak_measured'] = {'error': repr(e)}
        if env.world_size == 1 and not args.no_cpu_baseline:
            try:
                out['cpu_baseline'] = cpu_baseline(args.model, f_train)
            except Exception as e:
                out['cpu_baseline'] = {'error': repr(e)}
        print(json.dumps(out), flush=True)
    if env.distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
