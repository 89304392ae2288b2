"""CPU-only tests of the host side: the C-ABI library loads and exports every symbol of include/crl.h,
the reference plug-in surface (registry, cfg, counters, state_dict keys), arenas, schedule / scaler
logic, the data contract against the reference's own preprocess outputs, and the bucketed gradient
reducer on 2 gloo ranks.  No compute call is made (there is no GPU here)."""
import json
import os
import re
import random

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ------------------------------------------------------------------------------------------- C-ABI
def _header_decls():
    src = open(os.path.join(ROOT, 'include', 'crl.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    decls = {}
    for m in re.finditer(r'\b(?:int|size_t|const char\*)\s+(crl_\w+)\s*\(([^;]*?)\)\s*;', src, flags=re.S):
        args = m.group(2).strip()
        n = 0 if args in ('', 'void') else len([a for a in args.split(',') if a.strip()])
        decls[m.group(1)] = n
    return decls


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as ge
    from pixparse_amd import hip
    ge.build()
    decls = _header_decls()
    assert len(decls) >= 26
    lib = hip.load()
    for name, nargs in decls.items():
        assert hasattr(lib, name), f'{name} declared in crl.h but not exported'
        assert name in hip.SIGNATURES, f'{name} has no ctypes signature'
        assert len(hip.SIGNATURES[name][1]) == nargs, f'{name}: ctypes binds {len(hip.SIGNATURES[name][1])} args, header declares {nargs}'
    assert set(hip.SIGNATURES) == set(decls)
    assert hip.query('crl_version') == 1
    assert hip.query('crl_layernorm_bwd_ws_bytes', 1024) == 512 * 3 * 1024 * 4
    assert hip.query('crl_grad_norm_ws_bytes') > 0


def test_swin_window_attention_is_compiled_to_mfma():
    """north_star: the Swin window attention runs QK^T / PV on the matrix cores -- the gfx950 code hipcc emits for swin.hip holds
    v_mfma_f32_32x32x16_bf16 (8 + 8 forward, 40 backward per window and head) and the transposed LDS reads that feed them"""
    import subprocess, tempfile
    from pixparse_amd import build as b
    src = os.path.join(b.CSRC, 'swin.hip')
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, 'swin.s')
        r = subprocess.run([b._hipcc()] + b.FLAGS + ['--cuda-device-only', '-S', src, '-o', out], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-800:]
        asm = open(out).read()
    fwd = asm[asm.index('swin_attn_kernelILb0'):asm.index('swin_attn_kernelILb1')]
    bwd = asm[asm.index('swin_attn_kernelILb1'):]
    assert fwd.count('v_mfma_f32_32x32x16_bf16') >= 16 and bwd.count('v_mfma_f32_32x32x16_bf16') >= 40
    assert 'ds_read_b64_tr_b16' in fwd and 'ds_read_b64_tr_b16' in bwd


def test_gemm4w_streams_are_in_sync_and_counted():
    """csrc/gemm4w_body_{nt,nn,tn}.inc (the hand-placed main loops of the one-wave-per-SIMD GEMM, committed) are exactly what csrc/gen_gemm4w.py
    generates today; per K tile a wave issues the 128 MFMAs, the 32 fragment reads (64 half-reads for a k-major operand), the 16 LDS-DMA pieces and
    the two barriers the design counts; every LDS read is covered by a counted wait before its first consumer (the generator's own walk)."""
    import importlib.util, re
    from pixparse_amd import build as b
    spec = importlib.util.spec_from_file_location('gen_gemm4w', os.path.join(b.CSRC, 'gen_gemm4w.py'))
    gen = importlib.util.module_from_spec(spec)
    for k in ('G4W_DROP', 'G4W_OPTS'):
        assert not os.environ.get(k), f'{k} is set: the committed streams are the default build'
    spec.loader.exec_module(gen)
    for layout, (ka, kb) in gen.KINDS.items():
        stream, G = gen.generate(layout)
        text = gen.render(stream, G)
        assert text == open(os.path.join(b.CSRC, f'gemm4w_body_{layout}.inc')).read(), f'gemm4w_body_{layout}.inc is stale: run python pixparse_amd/csrc/gen_gemm4w.py'
        body = text[text.index('LOOP%=:'):text.index('EXIT%=:')]            # five K tiles (ring positions 1, 2, 3, 4, 0)
        assert body.count('v_mfma_f32_16x16x32_bf16') == 5 * 128 and body.count('s_barrier') == 5 * 2
        assert body.count('offen lds') == 5 * 16
        n_km = (ka == 'km') + (kb == 'km')
        assert body.count('ds_read_b128') == 5 * 16 * n_km and body.count('ds_read_b64_tr_b16') == 5 * 32 * (2 - n_km)
        assert len(re.findall(r'v_mfma_f32_16x16x32_bf16 a\[\d+:\d+\], v\[\d+:\d+\], v\[\d+:\d+\], 0\\n', text)) == 64     # the first k-step of an output tile starts the 64 accumulator tiles from C = 0
        assert not re.search(r'scratch_|v_accvgpr', text)
        # replay the stream: no MFMA may read a fragment register with an LDS read still pending (lgkmcnt is in-order)
        pending = []
        for ins in stream:
            if ins.kind == 'wait' and 'lgkmcnt' in ins.text:
                n = int(re.search(r'lgkmcnt\((\d+)\)', ins.text).group(1))
                pending = pending[len(pending) - n:] if n < len(pending) else pending
                if n == 0:
                    pending = []
            elif ins.kind == 'ds':
                pending.append(ins.writes)
            elif ins.kind == 'mfma':
                busy = set().union(*pending) if pending else set()
                assert not (busy & ins.reads), f'{layout}: {ins.text} reads a fragment that is still being loaded'
            elif ins.kind == 'label':
                pending = list(pending)


def test_scratch_grows_geometrically_once_frozen():
    """ops.Scratch after freeze_scratch(): a buffer that has to grow is retired (a captured graph may still write into it) and the new one is at
    least half as large again, so a rising sequence of requests retires a bounded total"""
    from pixparse_amd import ops
    was = ops.Scratch.frozen
    try:
        sc = ops.Scratch()
        dev = torch.device('cpu')
        sc.get(4 << 20, dev)
        ops.Scratch.frozen = True
        sizes = []
        for req in range(4 << 20, 16 << 20, 1 << 20):          # twelve requests, each 1 MiB larger
            sizes.append(sc.get(req + 1, dev).numel())
        assert len(sc.retired) <= 4 and sum(t.numel() for t in sc.retired) <= 2 * sizes[-1], (len(sc.retired), sizes)
        assert all(b >= a for a, b in zip(sizes, sizes[1:])) and sizes[-1] >= (15 << 20)
    finally:
        ops.Scratch.frozen = was


def test_attention_forward_streams_are_in_sync_and_counted():
    """csrc/attn_fwd4w_body.inc / attn_fwd2x_body.inc (the hand-placed attention forward at 512 / 256 registers per wave, committed) are exactly what
    csrc/gen_attn_fwd4w.py generates today; per 64-key tile a wave issues 32 large MFMAs, the 8 row-sum MFMAs, 64 v_exp, 32 conversions, 8 + 16 fragment
    reads, its 4 LDS-DMA pieces and one barrier; every LDS read is covered by a counted wait before its first consumer; a conversion never shares an MFMA
    gap with a v_exp that feeds it"""
    import importlib.util, re, sys
    from pixparse_amd import build as b
    for k in ('G4W_DROP', 'F4W_DROP', 'F4W_OPTS'):
        assert not os.environ.get(k), f'{k} is set: the committed streams are the default build'
    sys.path.insert(0, b.CSRC)
    try:
        spec = importlib.util.spec_from_file_location('gen_attn_fwd4w', os.path.join(b.CSRC, 'gen_attn_fwd4w.py'))
        gen = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(gen)
    finally:
        sys.path.remove(b.CSRC)
    for name, occ2 in zip(gen.FILES, (False, True)):
        stream = gen.generate(occ2)
        text = gen.render(stream, occ2)
        assert text == open(os.path.join(b.CSRC, name)).read(), f'{name} is stale: run python pixparse_amd/csrc/gen_attn_fwd4w.py'
        body = text[text.index('LOOP%=:'):text.index('s_branch LOOP%=')]            # four key tiles (ring positions 1, 2, 3, 0)
        assert body.count('v_mfma_f32_32x32x16_bf16') == 4 * 32 and body.count('v_mfma_f32_16x16x32_bf16') == 4 * 8
        assert body.count('v_exp_f32') == 4 * 64 and body.count('v_cvt_pk_bf16_f32') == 4 * 32
        assert body.count('ds_read_b128') == 4 * 8 and body.count('ds_read_b64_tr_b16') == 4 * 16
        assert body.count('offen lds') == 4 * 4 and body.count('s_barrier') == 4
        assert not re.search(r'scratch_', text)
        pending = []
        for ins in stream:
            if ins.kind == 'wait' and 'lgkmcnt' in ins.text:
                n = int(re.search(r'lgkmcnt\((\d+)\)', ins.text).group(1))
                pending = pending[len(pending) - n:] if 0 < n < len(pending) else ([] if n == 0 else pending)
            elif ins.kind == 'ds':
                pending.append(ins.writes)
            elif ins.kind == 'mfma':
                busy = set().union(*pending) if pending else set()
                assert not (busy & ins.reads), f'{name}: {ins.text} reads a fragment that is still being loaded'
    # the loop of both forms: a conversion reads only values exponentiated in an EARLIER gap (v_exp is a transcendental: its result is not forwarded)
    for name, occ2 in zip(gen.FILES, (False, True)):
        text = open(os.path.join(b.CSRC, name)).read()
        body = text[text.index('LOOP%=:'):text.index('s_branch LOOP%=')]
        fresh = set()
        for ln in body.splitlines():
            m = re.search(r'v_exp_f32 (v\d+),', ln)
            if 'v_mfma' in ln:
                fresh = set()
            elif m:
                fresh.add(m.group(1))
            elif 'v_cvt_pk_bf16_f32' in ln:
                a, c = re.search(r'v_cvt_pk_bf16_f32 v\d+, (v\d+), (v\d+)', ln).groups()
                assert a not in fresh and c not in fresh, f'{name}: {ln.strip()} reads a v_exp result of its own gap'


def test_attention_backward_stream_is_in_sync_and_its_hazard_pass_bites():
    """csrc/attn_bwd_sp_body.inc (the hand-placed single-pass attention backward, committed) is exactly what csrc/gen_attn_bwd_sp.py
    generates today; per 64-query tile pass it holds the 80 MFMAs the design counts, and the generator's hazard pass refuses a schedule
    that reads an MFMA result too early."""
    import importlib.util, re
    from pixparse_amd import build as b
    spec = importlib.util.spec_from_file_location('gen_attn_bwd_sp', os.path.join(b.CSRC, 'gen_attn_bwd_sp.py'))
    gen = importlib.util.module_from_spec(spec)
    for k in ('SPX_DROP', 'SPX_OPTS'):
        assert not os.environ.get(k), f'{k} is set: the committed stream is the default build'
    spec.loader.exec_module(gen)
    # round 6: ONE statement with two variants of the stream -- the general form and the form for the first key block of a chain (no running partial)
    gen.FIRST[0] = True
    first_stream, _ = gen.generate()
    gen.FIRST[0] = False
    stream, _ = gen.generate()
    text = gen.render(stream, first_stream)
    assert text == open(os.path.join(b.CSRC, 'attn_bwd_sp_body.inc')).read(), 'attn_bwd_sp_body.inc is stale: run python pixparse_amd/csrc/gen_attn_bwd_sp.py'
    body = text[text.index('"LOOP%=:'):text.index('"DRAIN0%=:')]
    fbody = text[text.index('"F_LOOP%=:'):text.index('"F_DRAIN0%=:')]
    for bd in (body, fbody):
        assert bd.count('v_mfma_f32_32x32x16_bf16') == 6 * 80 and bd.count('s_barrier') == 6          # six unrolled passes, one rendezvous each
        assert bd.count('v_exp_f32') == 6 * 64 and bd.count('ds_read_b64_tr_b16') == 6 * 64 and bd.count('ds_write_b64') == 6 * 16
    # the running-tile machinery per pass: 2 read-backs, 16 unpack operations, 2 LDS-DMA pieces of the partial -- all gone from the first-block variant
    assert body.count('ds_read2st64_b64') == 6 * 2 and fbody.count('ds_read2st64_b64') == 0
    assert body.count('%[rprev]') == 6 * 2 and fbody.count('%[rprev]') == 0
    assert body.count('s_waitcnt vmcnt(4) lgkmcnt(0)') == 6 and fbody.count('s_waitcnt vmcnt(2) lgkmcnt(0)') == 6
    assert not re.search(r'scratch_|v_pk_mul_f32', text)
    # the hazard pass: a VALU read of an accumulator right behind the MFMA that writes it must be refused
    H = gen.Hazards()
    H.emit(gen.mfma(('v', 0), ('v', 128), ('a', 128), ('v', 96), 16, 4, 4, 16))
    with pytest.raises(RuntimeError, match='read too early'):
        H.emit(gen.v_exp(0))
    # ... and an LDS read is waited for, with the count of younger LDS operations, before its first consumer
    H = gen.Hazards()
    H.emit(gen.ds_read_b128(128, 'arow0', 0))
    H.emit(gen.ds_read_b128(132, 'arow1', 0))
    H.emit(gen.mfma(('v', 0), ('v', 128), ('a', 128), '0', 16, 4, 4, 16))
    assert [i.text for i in H.out][2] == 's_waitcnt lgkmcnt(1)'


def test_binding_loads_torch_before_the_hip_library():
    """regression: dlopen of libcruller_hip.so before torch maps a second HIP runtime (torch ships its own libamdhip64) and
    every later launch fails with "no ROCm-capable device" -- build() followed by smoke() in one process hit this"""
    import subprocess, sys
    code = "import sys; import pixparse_amd.hip as h; assert 'torch' in sys.modules; h.load(); print('ok')"
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0 and r.stdout.strip().endswith('ok'), r.stderr[-500:]


def test_argument_validation_without_gpu():
    """error paths that return before any launch: wrong shapes come back as HipLibraryError with the text"""
    from pixparse_amd import hip
    hip.load()
    with pytest.raises(hip.HipLibraryError, match='multiple of 32'):
        hip.call('crl_gemm_bf16', hip.NT, hip.EPI_BF16, 8, 8, 40, 16, 40, 16, 40, None, 16, 8, None, 0, None, 0, 1.0, 0, None, 0, None)
    with pytest.raises(hip.HipLibraryError, match='empty'):
        hip.call('crl_attn_fwd', 16, 64, 64, 16, 64, 64, 16, 64, 64, 16, 64, 64, 16, 0, 1, 4, 4, 0.125, 0, 0, 0.0, 0, 0, 0, None)
    with pytest.raises(hip.HipLibraryError, match='window'):
        hip.call('crl_swin_attn_fwd', 16, 16, 16, 1, 18, 18, 1, 9, 0, 0.1, None)


# ------------------------------------------------------------------------------------------- model surface
def test_decode_entry_points_validate_arguments_without_gpu():
    """the generation kernels refuse bad shapes / null pointers before touching the device"""
    from pixparse_amd import hip
    lib = hip.load()
    assert lib.crl_linear_skinny_bf16(0, 17, 64, 64, 16, 64, 16, 64, None, 16, 64, None, 0, None, 0, None) != 0
    assert 'M = 17' in hip.last_error()
    assert lib.crl_linear_skinny_bf16(0, 4, 64, 60, 16, 64, 16, 64, None, 16, 64, None, 0, None, 0, None) != 0      # K % 8
    assert lib.crl_linear_skinny_bf16(5, 4, 64, 64, 16, 64, 16, 64, None, 16, 64, None, 0, None, 0, None) != 0      # F32_ACC epilogue
    assert 'epilogue' in hip.last_error()
    assert lib.crl_linear_skinny_bf16(0, 4, 64, 64, None, 64, 16, 64, None, 16, 64, None, 0, None, 0, None) != 0
    assert lib.crl_attn_decode(None, 64, 16, 64, 64, 16, 64, 64, 16, 64, 1, 1, 8, 0.125, None, None, 0, 16, 1024, None) != 0
    assert 'null' in hip.last_error()
    assert lib.crl_attn_decode(16, 64, 16, 64, 64, 16, 64, 64, 16, 64, 1, 1, 8, 0.125, None, None, 0, 16, 8, None) != 0   # workspace too small
    assert 'workspace' in hip.last_error()
    assert hip.query('crl_attn_decode_ws_bytes', 8, 16, 6189) == 8 * 16 * 7 * 66 * 4      # 128 (b, h) x 7 splits of 1024 keys
    assert lib.crl_prof_end(6, None, None, None) != 0 and 'not profiling' in hip.last_error()


def test_state_dict_keys_match_reference_checkpoint_layout():
    from oracle import ref_cpu as R
    from pixparse_amd.models import Cruller, get_model_config
    cfg = get_model_config('cruller_small')
    model = Cruller(cfg, vocab_size=1027)
    spec = R.ModelSpec('swin_tiny_patch4_window7_224', 'facebook/bart-base', 2, 128, (224, 224), 3, vocab=1027)
    sd = model.state_dict()
    want = spec.param_shapes()
    got = {k: tuple(v.shape) for k, v in sd.items() if not k.endswith('lm_head.weight')}
    assert got == want
    assert sd['text_decoder.trunk.lm_head.weight'].data_ptr() == sd['text_decoder.trunk.model.decoder.embed_tokens.weight'].data_ptr()
    assert model.image_encoder.trunk.pretrained_cfg['mean'] == (0.485, 0.456, 0.406)
    # parameters are views of ONE flat arena, 16-byte aligned
    base = model.arena.p.data_ptr()
    for n, p in model.named_parameters():
        assert base <= p.data_ptr() < base + model.arena.total * 4 and (p.data_ptr() - base) % 256 == 0
    # checkpoint interchange: load an oracle-initialised state dict by name
    params = R.init_params(spec, seed=3)
    params['text_decoder.trunk.lm_head.weight'] = params['text_decoder.trunk.model.decoder.embed_tokens.weight']
    model.load_state_dict(params)
    assert torch.equal(model.arena.param('image_encoder.trunk.norm.weight'), params['image_encoder.trunk.norm.weight'])


@pytest.mark.parametrize('name,enc,dec,layers,L,img,ch', [
    ('cruller_base', 'vit_base_patch16_224', 'facebook/bart-base', 4, 1024, (576, 448), 1),
    ('cruller_large_1280x960', 'vit_large_patch14_clip_224.datacompxl', 'facebook/bart-large', 10, 1024, (1280, 960), 3)])
def test_large_config_layouts_without_materialising(name, enc, dec, layers, L, img, ch):
    from oracle import ref_cpu as R
    from pixparse_amd.layers.engines import BartEngine, ViTEngine
    from pixparse_amd.models import get_model_config
    from pixparse_amd.models.archs import BART_ARCHS, VIT_ARCHS
    cfg = get_model_config(name)
    assert cfg.image_encoder.name == enc and cfg.text_decoder.num_decoder_layers == layers and tuple(cfg.image_encoder.image_size) == img
    spec = R.ModelSpec(enc, dec, layers, L, img, ch, vocab=50267)
    mine = {('image_encoder.trunk.' + i[0]): tuple(i[1]) for i in ViTEngine.param_shapes(VIT_ARCHS[enc], ch, img)}
    mine.update({('text_decoder.trunk.' + i[0]): tuple(i[1]) for i in BartEngine.param_shapes(BART_ARCHS[dec], layers, 50267, L)})
    assert mine == spec.param_shapes()
    n = sum(int(torch.tensor(s).prod()) for s in mine.values())
    if name.startswith('cruller_large'):
        assert abs(n - 529.7e6) < 0.5e6          # 529.7 M parameters


def test_no_cpu_fallback():
    from pixparse_amd.framework import DeviceEnv
    from pixparse_amd.models import Cruller, get_model_config
    from pixparse_amd.task import TaskFactory
    model = Cruller(get_model_config('cruller_small'), vocab_size=515)
    with pytest.raises(RuntimeError, match='no CPU path'):
        model(torch.zeros(1, 3, 224, 224), torch.zeros(1, 8, dtype=torch.int64))
    with pytest.raises(TypeError):
        model.to(torch.bfloat16)
    with pytest.raises(ValueError):
        TaskFactory.create_task('nope', {}, DeviceEnv('cpu'), None)
    with pytest.raises(NotImplementedError):
        TaskFactory.create_task('donut_eval_ocr', {}, DeviceEnv('cpu'), None)   # in the reference's registry, not built here
    task, _ = TaskFactory.create_task('cruller_eval_docvqa', dict(model=get_model_config('cruller_small')), DeviceEnv('cpu'), None)
    with pytest.raises(RuntimeError, match='no CPU path'):
        task.setup()                                                             # eval tasks refuse a CPU device as loudly as the train task


def test_decoder_dropout_switch_cpu():
    """dropout is an explicit opt-in (task cfg `decoder_dropout`, Cruller.set_train_dropout).  With it on, a model gets
    every train-mode regulariser of its reference counterpart: hidden-state dropout, bart-base's attention-probability and activation
    dropout, and the Swin encoder's drop-path (timm default 0.1)"""
    from pixparse_amd.models import Cruller, get_model_config
    from pixparse_amd.task import TaskCrullerPretrainCfg
    assert TaskCrullerPretrainCfg().decoder_dropout is False
    small = Cruller(get_model_config('cruller_small'), vocab_size=515)        # swin_tiny + bart-base
    small.set_train_dropout(True, seed=7)
    assert small._drop == (0.1, 7, 0.1, 0.1, 0.1)                              # (hidden, seed, attention, activation, drop-path)
    small.set_train_dropout(False)
    assert small._drop is None
    large = Cruller(get_model_config('cruller_base'), vocab_size=515)         # ViT (no drop-path) + bart-base
    large.set_train_dropout(True)
    assert large._drop == (0.1, 0, 0.1, 0.1, 0.0)


def test_task_surface_and_counters_cpu():
    """constructor side of the plug-in (runs without a GPU); train_setup refuses a CPU device loudly"""
    from types import SimpleNamespace
    from pixparse_amd.framework import DeviceEnv, OptimizationCfg
    from pixparse_amd.models import ImageEncoderCfg, ModelCfg, TextDecoderCfg
    from pixparse_amd.models.archs import register_arch
    from pixparse_amd.task import TaskCrullerPretrain, TaskFactory
    register_arch('vit', 'vit_cpu_t', dict(patch=8, dim=64, depth=1, heads=1, mlp_ratio=2, ln_eps=1e-6, pre_norm=False, mean=(0.5,) * 3, std=(0.5,) * 3))
    register_arch('bart', 'bart_cpu_t', dict(d_model=64, heads=1, ffn=128, ln_eps=1e-5, vocab=50265, dropout=0.0))
    mcfg = ModelCfg(ImageEncoderCfg('vit_cpu_t', 'L', (32, 24), False), TextDecoderCfg('bart_cpu_t', False, 1, 16))
    args = SimpleNamespace(num_intervals=3, num_warmup_intervals=1, eval_frequency=7, opt=OptimizationCfg(grad_accum_steps=2),
                           dtype='bfloat16', amp=True, model_name=None, model=mcfg)
    task, cfg = TaskFactory.create_task('cruller_pretrain', args, DeviceEnv('cpu'), None)
    assert isinstance(task, TaskCrullerPretrain) and cfg.model_name == 'custom'
    assert (task.step, task.batch_idx, task.interval_idx, task.interval_batch_idx, task.start_interval) == (0, 0, 0, 0, 0)
    assert task.num_intervals == 3 and task.eval_frequency == 7 and task.vocab_size == 50267
    assert task.model.state_dict()['text_decoder.trunk.model.decoder.embed_tokens.weight'].shape == (50267, 64)
    assert task.num_image_chs == 1 and abs(task.img_mean - 0.5) < 1e-12
    x = task.image_preprocess_train(torch.rand(1, 40, 30))
    assert x.shape == (1, 32, 24)
    with pytest.raises(RuntimeError, match='MI355X'):
        task.train_setup(10)
    with pytest.raises(NotImplementedError):
        TaskFactory.create_task('cruller_pretrain', SimpleNamespace(**{**vars(args), 'dtype': 'float16'}), DeviceEnv('cpu'), None)


# ------------------------------------------------------------------------------------------- schedule / scaler / data
def test_cosine_schedule_matches_golden(golden_dir):
    from pixparse_amd.framework.optim import CosineLRScheduler
    meta = json.load(open(os.path.join(golden_dir, 'g5_optim.json')))['sched']

    class Opt:
        param_groups = [dict(lr=meta['base'], initial_lr=meta['base'])]
    s = CosineLRScheduler(Opt, t_initial=meta['t_initial'], warmup_t=meta['warmup_t'])
    for t, lr in meta['lrs'].items():
        s.step_update(int(t))
        assert abs(Opt.param_groups[0]['lr'] - lr) < 1e-15


def test_loss_scaler_bookkeeping():
    """host side of the device-resident GradScaler: defaults, attach() moves scale / tracker into words 4 / 5 of the optimiser
    state vector, state_dict round trip (the update rule itself runs in crl_grad_norm_scaled: tests/test_kernels_gpu.py)"""
    from pixparse_amd.framework.optim import STATE_FLOATS, LossScaler
    sc = LossScaler(growth_interval=3)
    assert sc.get_scale() == 65536.0 and sc.scale_tensor() is None and sc.growth_interval == 3
    st = torch.zeros(STATE_FLOATS)
    sc.attach(st)
    assert st[4] == 65536.0 and st[5] == 0.0 and sc.scale_tensor().data_ptr() == st[4:5].data_ptr()
    st[4], st[5] = 1024.0, 2.0            # what the device kernel would leave after backoffs
    sd = sc.state_dict()
    assert sd['scale'] == 1024.0 and sd['_growth_tracker'] == 2 and sd['growth_interval'] == 3
    sc2 = LossScaler().attach(torch.zeros(STATE_FLOATS))
    sc2.load_state_dict(sd)
    assert sc2.get_scale() == 1024.0 and sc2.get_growth_tracker() == 2
    off = LossScaler(enabled=False)
    assert off.get_scale() == 1.0 and off.attach(torch.zeros(STATE_FLOATS)).scale_tensor() is None


class _StubTok:
    """the tokenizer tests/golden/make_golden.py used to run the reference's preprocess.py"""
    pad_token_id, eos_token = 1, '</s>'
    specials = {'</s>': 2, '<s_pretrain>': 50266, '<sep/>': 50265}

    def convert_tokens_to_ids(self, t):
        return self.specials[t]

    def __call__(self, text, add_special_tokens=False, return_tensors='pt', max_length=None, padding='max_length', truncation=True):
        ids, i = [], 0
        while i < len(text):
            for s, sid in self.specials.items():
                if text.startswith(s, i):
                    ids.append(sid)
                    i += len(s)
                    break
            else:
                ids.append(ord(text[i]) + 100)
                i += 1
        ids = (ids[:max_length] + [1] * max_length)[:max_length]
        return type('E', (), {'input_ids': torch.tensor([ids])})()


def test_preprocess_matches_reference_outputs(golden_dir):
    from pixparse_amd.data import preprocess_ocr_anno, preprocess_text_anno
    cases = json.load(open(os.path.join(golden_dir, 'g3_preprocess.json')))
    tok = _StubTok()
    for c in cases:
        if c['fn'] == 'ocr':
            out, meta = preprocess_ocr_anno(c['anno'], tok, c['L'], '<s_pretrain>', '<s_pretrain>', generator=random.Random(0))
            assert meta == c['meta']
        else:
            out = preprocess_text_anno(c['anno'], tok, c['L'], '<s_pretrain>', '<s_pretrain>')
        assert out['text'][0].tolist() == c['text'] and out['target'][0].tolist() == c['target']
    with pytest.raises(RuntimeError):
        preprocess_ocr_anno({'pages': []}, tok, 8, '<s_pretrain>', '<s_pretrain>', generator=random.Random(0))
    with pytest.raises(RuntimeError):
        preprocess_ocr_anno({'pages': [{'text': []}, {'text': []}]}, tok, 8, '<s_pretrain>', '<s_pretrain>', generator=random.Random(0))


def test_synthetic_loader_contract():
    from pixparse_amd.data import SyntheticLoaderBundle
    lb = SyntheticLoaderBundle(batch_size=2, num_batches=3, in_chans=1, img_size=(32, 24), max_length=16, vocab_size=50267)
    lb.set_interval(0)
    batches = list(lb.loader)
    assert len(batches) == 3 == lb.num_batches and lb.num_samples == 6
    img, text, tgt = batches[0]
    assert img.shape == (2, 1, 32, 24) and text.shape == (2, 16) and text.dtype == torch.int64
    assert (text[:, 0] == 50266).all() and (text[:, -1] == 2).all() and (tgt[:, 0] == -100).all() and (tgt[:, 1:] == text[:, 1:]).all()


def test_byte_tokenizer_roundtrip():
    from pixparse_amd.tokenizers import ByteBartTokenizer
    t = ByteBartTokenizer()
    assert t.add_special_tokens({'additional_special_tokens': sorted({'<sep/>', '<s_pretrain>'})}) == 2 and len(t) == 50267
    ids = t('<s_pretrain>héllo</s>', max_length=16).input_ids[0]
    assert ids[0] == t.convert_tokens_to_ids('<s_pretrain>') and 2 in ids.tolist() and ids[-1] == 1
    assert t.decode(ids, skip_special_tokens=True) == 'héllo'


# ------------------------------------------------------------------------------------------- arena + reducer (gloo, 2 ranks)
def _make_arena():
    from pixparse_amd.layers.arena import ParamArena
    from pixparse_amd.layers.engines import BartEngine
    a = ParamArena()
    a.add('enc.w', (33, 7))
    for item in BartEngine.param_shapes(dict(d_model=64, heads=1, ffn=128, ln_eps=1e-5), 2, 131, 16):
        a.add(item[0], item[1], item[2] if len(item) > 2 else None)
    a.materialize('cpu')
    return a


def test_arena_layout_and_fused_views():
    from pixparse_amd.layers.engines import BartEngine, Buffers
    a = _make_arena()
    a.alloc_training_state()
    offs = [e.offset for e in a.entries.values()]
    assert offs == sorted(offs) and all(o % 64 == 0 for o in offs)
    e = a.entries['model.decoder.embed_tokens.weight']
    assert e.alloc == 256 * 64 and e.numel == 131 * 64            # vocab rows padded to a multiple of 128
    eng = BartEngine(dict(d_model=64, heads=1, ffn=128, ln_eps=1e-5), 2, 131, 16, a, '', Buffers('cpu'))
    a.pb = torch.zeros(a.total, dtype=torch.bfloat16)
    lp = 'model.decoder.layers.1.'
    w = eng.fw('p', lp, 'self_attn', 'q_proj', 3)
    a.param(lp + 'self_attn.k_proj.weight').fill_(2.0)
    a.param(lp + 'self_attn.v_proj.bias').fill_(3.0)
    assert w.shape == (192, 64) and float(w[64:128].min()) == 2.0 and float(w[:64].abs().max()) == 0 and float(w[128:].abs().max()) == 0
    assert float(eng.fb('p', lp, 'self_attn', 'q_proj', 3)[128:].min()) == 3.0
    kv = eng.fw('g', lp, 'encoder_attn', 'k_proj', 2)
    assert kv.data_ptr() == a.grad(lp + 'encoder_attn.k_proj.weight').data_ptr() and kv.shape == (128, 64)


def _reducer_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from pixparse_amd.framework import DeviceEnv
    from pixparse_amd.framework.reducer import BucketedGradReducer
    env = DeviceEnv('cpu')
    assert env.world_size == world and env.global_rank == rank
    a = _make_arena()
    a.alloc_training_state()
    torch.manual_seed(100 + rank)
    a.p.normal_()
    red = BucketedGradReducer(a, world, bucket_bytes=16 << 10)
    red.broadcast_params(0)
    p_sum = a.p.clone()
    dist.all_reduce(p_sum)
    ok = bool(torch.allclose(p_sum, a.p * world))
    # buckets tile the arena exactly once, last bucket first
    cover = torch.zeros(a.total)
    for s, e in red.buckets:
        cover[s:e] += 1
    ok &= bool((cover == 1).all()) and red.buckets[0][1] == a.total and len(red.buckets) > 3
    g_local = torch.randn(a.total, generator=torch.Generator().manual_seed(7 + rank))
    a.g.copy_(g_local)
    expect = sum(torch.randn(a.total, generator=torch.Generator().manual_seed(7 + r)) for r in range(world))
    names = list(a.entries)
    red.begin()
    fired = []
    for name in reversed(names):            # the backward sweep reports entries from the end of the arena
        red.on_ready(name)
        fired.append(red._next)
    ok &= fired == sorted(fired) and 0 < fired[len(fired) // 2] < len(red.buckets)   # overlap: buckets fire progressively
    red.finish()
    ok &= bool(torch.allclose(a.g, expect, atol=1e-5)) and red.grad_divisor() == world
    # no_sync micro-step: nothing is exchanged
    a.g.copy_(g_local)
    red.enabled = False
    red.begin()
    for name in reversed(names):
        red.on_ready(name)
    red.finish()
    ok &= bool(torch.equal(a.g, g_local)) and red.grad_divisor() == 1.0
    obj = env.broadcast_object({'date': 'x'} if rank == 0 else None)
    ok &= obj == {'date': 'x'} and len(env.all_gather_object(rank)) == world
    q.put((rank, ok))
    dist.destroy_process_group()


def test_reducer_reserves_cus_for_rccl_by_itself(monkeypatch):
    """world_size > 1 switches the CU reservation of the persistent GEMMs on without anybody setting a variable: NCCL_MAX_NCHANNELS CUs
    when RCCL's channel count is pinned, else 16; PIXPARSE_AMD_RCCL_CUS overrides (0 = off); one rank reserves nothing.  stats() carries
    the bucket geometry bench.py prints."""
    from types import SimpleNamespace
    from pixparse_amd.framework.reducer import BucketedGradReducer

    def make(world, active=None):
        arena = SimpleNamespace(total=50_000_000, p=SimpleNamespace(is_cuda=False), g=None, entries={})
        r = BucketedGradReducer(arena, world, active=active)
        return r
    for k in ('PIXPARSE_AMD_RCCL_CUS', 'NCCL_MAX_NCHANNELS'):
        monkeypatch.delenv(k, raising=False)
    assert make(8)._auto_reserved_cus() == 16 and make(1)._auto_reserved_cus() == 0 and make(1, active=True)._auto_reserved_cus() == 0
    monkeypatch.setenv('NCCL_MAX_NCHANNELS', '32')
    assert make(8)._auto_reserved_cus() == 32
    monkeypatch.setenv('PIXPARSE_AMD_RCCL_CUS', '0')
    assert make(8)._auto_reserved_cus() == 0
    monkeypatch.setenv('PIXPARSE_AMD_RCCL_CUS', '24')
    assert make(8)._auto_reserved_cus() == 24 and make(1)._auto_reserved_cus() == 24
    r = make(8)
    assert r.reserved_cus == 0                    # a CPU arena never touches the GEMM launch geometry
    st = r.stats()
    assert st['buckets'] == 3 and st['bucket_bytes'] == 64 << 20 and st['comm_exposed_ms'] == 0.0 and st['reductions'] == 0


def test_attention_backward_chain_length_follows_the_available_cus():
    """crl_attn_bwd_chain_for: key blocks per workgroup of the single-pass attention backward = argmin of a simulated makespan (longest
    workgroups first on the CUs not reserved for RCCL), stretched by 0.6 % per link, + 0.1 per slab (at 128 heads) + 0.03 per link.  Host arithmetic only.  cfg-3 (25 key blocks, 128
    heads): 768 chains of 4 fill 256 CUs three times and the 128 one-block remainders half a round (makespan 13, as without chains, 7 slabs
    instead of 25); with 16 CUs set aside that shape needs a fourth round, chains of 2 do not; one key block cannot be chained; a forced
    length is clamped to the key blocks there are; few heads on many CUs are not worth chaining (every link serialises work)."""
    from pixparse_amd import hip
    q = lambda nk, bh: hip.query('crl_attn_bwd_chain_for', nk, bh)
    try:
        assert q(6189, 128) == 4
        assert q(200, 128) == 1 and q(256, 8) == 1
        assert q(6189, 2) == 1                                        # 50 workgroups on 256 CUs: a chain only makes the longest one longer
        c5 = q(24935, 32)                                             # cfg-5: 98 key blocks, 32 heads
        assert 2 <= c5 <= 98 and -(-98 // c5) < 98
        hip.call('crl_gemm_set_reserved_cus', 16)
        assert q(6189, 128) == 2
        hip.call('crl_gemm_set_reserved_cus', 0)
        hip.call('crl_attn_bwd_set_chain', 7)
        assert q(6189, 128) == 7 and q(300, 128) == 2
        with pytest.raises(hip.HipLibraryError):
            hip.call('crl_attn_bwd_set_chain', -1)
        assert hip.query('crl_attn_bwd_chain_for', 0, 4) == -1
    finally:
        hip.call('crl_gemm_set_reserved_cus', 0)
        hip.call('crl_attn_bwd_set_chain', 0)
    assert q(6189, 128) == 4


def test_attention_backward_query_split_follows_the_simulated_makespan():
    """crl_attn_bwd_qsplit_for: the key blocks a head has left over after its full chains are split between two workgroups (by query halves) when that
    shortens the simulated makespan by at least half a key block.  cfg-3 (25 key blocks, 128 heads, chains of 4): 768 chains fill 256 CUs three times,
    the 128 one-block remainders would give half the CUs a 13th block -- 256 half blocks give every CU 12.5: split.  cfg-2 (10 key blocks, 96 heads,
    chains of 4 + a remainder of 2): the remainders already fit beside the last full chains, nothing to gain.  No remainder, a forced chain (then only
    with qsplit forced too) and crl_attn_bwd_set_qsplit(0): no split.  Host arithmetic only."""
    from pixparse_amd import hip
    s = lambda nk, bh: hip.query('crl_attn_bwd_qsplit_for', nk, bh)
    try:
        assert s(6189, 128) == 1
        assert s(2401, 96) == 0
        assert s(1024, 128) == 0                                      # four key blocks, one chain of four: no remainder
        hip.call('crl_attn_bwd_set_qsplit', 0)
        assert s(6189, 128) == 0
        hip.call('crl_attn_bwd_set_qsplit', -1)
        hip.call('crl_attn_bwd_set_chain', 4)                         # forced chain: the automatic split is off ...
        assert s(6189, 128) == 0
        hip.call('crl_attn_bwd_set_qsplit', 1)                        # ... unless forced too
        assert s(6189, 128) == 1 and s(1024, 128) == 0
        with pytest.raises(hip.HipLibraryError):
            hip.call('crl_attn_bwd_set_qsplit', 2)
        assert hip.query('crl_attn_bwd_qsplit_for', 0, 4) == -1
    finally:
        hip.call('crl_attn_bwd_set_qsplit', -1)
        hip.call('crl_attn_bwd_set_chain', 0)
    assert s(6189, 128) == 1


def test_bucketed_reducer_two_gloo_ranks():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + random.randint(0, 2000)
    procs = [ctx.Process(target=_reducer_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(60)
    assert sorted(res) == [(0, True), (1, True)]


def test_app_flag_surface():
    """reference flag spelling (README.md:19-40): nested dataclass flags with dash or underscore variants"""
    from pixparse_amd.app.train import parse_args
    t, k, d = parse_args(['--task.model-name', 'cruller_small', '--task.dtype', 'bfloat16', '--task.opt.learning-rate', '3e-4',
                          '--task.opt.betas', '0.9 0.98', '--task.opt.clip_grad_value', '1.0', '--task.opt.clip-grad-mode', 'norm',
                          '--task.opt.grad-accum-steps', '4', '--data.train.batch-size', '2', '--train.num-intervals', '3',
                          '--train.output-dir', '/tmp/x'])
    assert k.model_name == 'cruller_small' and k.model.image_encoder.name == 'swin_tiny_patch4_window7_224'
    assert k.opt.learning_rate == 3e-4 and k.opt.betas == (0.9, 0.98) and k.opt.clip_grad_value == 1.0 and k.opt.grad_accum_steps == 4
    assert t.num_intervals == 3 and t.output_dir == '/tmp/x' and d.batch_size == 2 and k.dtype == 'bfloat16'


@pytest.mark.parametrize('i,o', [(1754, 1280), (1240, 960), (100, 37), (50, 64), (224, 224)])
def test_aa_bicubic_tables_match_torch(i, o):
    """filter tables of the GPU preprocess kernel == aten upsample_bicubic2d_aa (what torchvision Resize(BICUBIC, antialias) runs)"""
    import torch.nn.functional as F
    from pixparse_amd.data.gpu_preprocess import aa_bicubic_tables
    xmin, xsize, w = aa_bicubic_tables(i, o)
    x = torch.rand(1, 1, 1, i, generator=torch.Generator().manual_seed(0))
    ref = F.interpolate(x, size=(1, o), mode='bicubic', antialias=True, align_corners=False)[0, 0, 0]
    got = torch.stack([(w[k, :xsize[k]] * x[0, 0, 0, xmin[k]:xmin[k] + xsize[k]]).sum() for k in range(o)])
    assert float((got - ref).abs().max()) < 2e-6
    assert int((xmin + xsize).max()) <= i and float((w.sum(1) - 1).abs().max()) < 1e-5


# ------------------------------------------------------------------------------------------- f-3: fine-tune glue
def test_json2token_and_token2json_match_reference_outputs(golden_dir):
    """G6: outputs of the reference's utils/json_utils.py json2token / token2json (generated by make_golden.gen_finetune)"""
    from pixparse_amd.utils import json2token, token2json
    g = json.load(open(os.path.join(golden_dir, 'g6_finetune.json')))
    for c in g['json2token']:
        out = json2token(c['obj'], c['specials'], [], True, c['sort_json_key'])
        text = out if isinstance(out, str) else out[0]
        toks = [] if isinstance(out, str) else sorted(out[1])
        assert text == c['text'], c['obj']
        assert toks == c['key_tokens']
        assert token2json(text, added_vocab={t: i for i, t in enumerate(c['specials'])}) == c['token2json']
    # the reference's mutable default argument is not reproduced: a second call starts clean
    assert json2token({'a': 'b'}, [])[1] and sorted(json2token({'c': 'd'}, [])[1]) == ['</s_c>', '<s_c>']


def _finetune_task(name):
    from pixparse_amd.framework import DeviceEnv
    from pixparse_amd.models import ImageEncoderCfg, ModelCfg, TextDecoderCfg
    from pixparse_amd.models.archs import register_arch
    from pixparse_amd.task import TaskFactory
    register_arch('vit', 'vit_ft_test', dict(patch=8, dim=128, depth=1, heads=2, mlp_ratio=4, ln_eps=1e-6, pre_norm=False, mean=(0.5,) * 3, std=(0.25,) * 3))
    register_arch('bart', 'bart_ft_test', dict(d_model=128, heads=2, ffn=256, ln_eps=1e-5, vocab=509, dropout=0.0))
    model = ModelCfg(image_encoder=ImageEncoderCfg(name='vit_ft_test', image_fmt='L', image_size=(32, 40), pretrained=False),
                     text_decoder=TextDecoderCfg(name='bart_ft_test', pretrained=False, num_decoder_layers=1, max_length=512))
    task, cfg = TaskFactory.create_task(name, dict(model=model, dtype='bfloat16'), DeviceEnv(init_device_type='cpu'), None)
    return task


def test_finetune_prompt_masking_matches_reference(golden_dir):
    """G6: text_input_to_target of the three reference fine-tune tasks on the same token ids"""
    g = json.load(open(os.path.join(golden_dir, 'g6_finetune.json')))
    names = {'TaskCrullerFinetuneRVLCDIP': 'cruller_finetune_rvlcdip', 'TaskCrullerFinetuneDOCVQA': 'cruller_finetune_docvqa',
             'TaskCrullerFinetuneCORD': 'cruller_finetune_cord'}
    for c in g['text_input_to_target']:
        task = _finetune_task(names[c['task']])
        task.tokenizer.trunk.added = dict(g['added_tokens'])      # the ids the fixture was generated with
        task.prompt_end_token = c['prompt_end']
        got = task.text_input_to_target(torch.tensor(c['ids']))
        assert got.tolist() == c['target'], c['text']


def test_finetune_collate_and_token_staging():
    """tokens are added in two stages and the embedding table follows (ref RVLCDIP :147-162, :225-236); collate returns
    decoder inputs / labels shifted against each other with prompt and pads masked"""
    import numpy as np
    from pixparse_amd.task import TaskCrullerFinetuneCORD, TaskCrullerFinetuneDOCVQA, TaskCrullerFinetuneRVLCDIP
    t = _finetune_task('cruller_finetune_rvlcdip')
    assert isinstance(t, TaskCrullerFinetuneRVLCDIP) and t.vocab_size == 50267 and t.model.vocab_size == 50267
    pre = {k: v.clone() for k, v in t.model.state_dict().items()}
    t.state_dict = {'module.' + k: v for k, v in pre.items()}      # what app/train.py does on --resume (ref app/train.py:156)
    t.resume = True
    with pytest.raises(RuntimeError, match='MI355X'):              # tokens / resize / checkpoint happen before the device check
        t.train_setup(4)
    assert t.newly_added_num == 19 and t.vocab_size == 50267 + 19 == t.model.vocab_size   # <s_rvlcdip>, <s_class>, </s_class>, 16 classes
    emb = t.model.state_dict()['text_decoder.trunk.model.decoder.embed_tokens.weight']
    assert emb.shape[0] == 50286 and torch.equal(emb[:50267], pre['text_decoder.trunk.model.decoder.embed_tokens.weight'])
    assert callable(t.state_dict)                                   # the method is back after the checkpoint was consumed
    img = (np.arange(50 * 60 * 3) % 251).astype(np.uint8).reshape(50, 60, 3)
    b = t.collate_fn([{'image': img, 'label': 0}, {'image': img, 'label': 15}])
    ids = t.tokenizer.trunk.convert_tokens_to_ids
    assert b['image'].shape == (2, 1, 32, 40) and b['label'].shape == (2, 4) and b['text_target'].shape == (2, 4)
    assert b['label'][0].tolist() == [ids('<s_rvlcdip>'), ids('<letter/>'), 2, 1]
    assert b['text_target'][1].tolist() == [ids('<memo/>'), 2, -100, -100]
    # grayscale + normalise like torchvision: ToTensor -> Grayscale -> Resize(bicubic, antialias) -> Normalize(mean 0.5, std 0.25)
    x = torch.from_numpy(img).permute(2, 0, 1).float() / 255
    gray = (0.2989 * x[0] + 0.587 * x[1] + 0.114 * x[2])[None, None]
    ref = (torch.nn.functional.interpolate(gray, size=(32, 40), mode='bicubic', antialias=True, align_corners=False)[0] - 0.5) / 0.25
    assert torch.allclose(b['image'][0], ref, atol=1e-6)

    d = _finetune_task('cruller_finetune_docvqa')
    assert isinstance(d, TaskCrullerFinetuneDOCVQA)
    d.tokenizer.trunk.add_special_tokens({'additional_special_tokens': sorted(set(d.special_tokens_finetune))})
    qa = '<s_question>who?</s_question><s_answer>me</s_answer>'
    b = d.collate_fn([{'image': img, 'labels': [qa]}])
    assert b['label'].shape == (1, 511)
    tgt = b['text_target'][0]
    n_valid = int((tgt != -100).sum())
    assert n_valid == 4 and tgt[tgt != -100].tolist() == [4 + ord('m'), 4 + ord('e'), d.tokenizer.trunk.convert_tokens_to_ids('</s_answer>'), 2]

    c = _finetune_task('cruller_finetune_cord')
    assert isinstance(c, TaskCrullerFinetuneCORD) and len(c.special_tokens_finetune) == 3 + 54
    c.tokenizer.trunk.add_special_tokens({'additional_special_tokens': sorted(set(c.special_tokens_finetune))})
    gt = repr({'gt_parse': {'menu': [{'nm': 'tea', 'cnt': '1'}], 'total': {'total_price': '3'}}})
    seq = c._sequence_for({'ground_truth': gt})
    assert seq == '<s_cord><s_menu><s_nm>tea</s_nm><s_cnt>1</s_cnt></s_menu><s_total><s_total_price>3</s_total_price></s_total></s>'
    b = c.collate_fn([{'image': img, 'ground_truth': gt}])
    assert b['label'][0, 0] == c.tokenizer.trunk.convert_tokens_to_ids('<s_cord>') and b['text_target'][0, 0] == c.tokenizer.trunk.convert_tokens_to_ids('<s_menu>')


# ------------------------------------------------------------------------------------------- f-4: OCR metrics
def test_cer_wer_definitions():
    """jiwer's cer / wer as the reference calls them (utils/ocr_utils.py:32-46, :111-140): batch-summed edit distance over
    batch-summed reference length; <pad> words removed; WER collapses repeated spaces, CER does not. jiwer itself is not
    installable here, so these are known answers worked out from its documented definitions (parity unpinned for jiwer)."""
    from pixparse_amd.utils.ocr_utils import _cer_tokens, _edit_distance, _wer_tokens, get_cer_wer_metrics
    assert _edit_distance('kitten', 'sitting') == 3 and _edit_distance([], ['a']) == 1 and _edit_distance('abc', 'abc') == 0
    assert _wer_tokens('  a   b <pad> c<pad> ') == ['a', 'b', 'c<pad>'] and _cer_tokens(' ab <pad> ') == ['a', 'b']
    m = get_cer_wer_metrics(None, None, {}, ['the cat sat', 'hello  wrld <pad> <pad>'], ['the cat sat on', 'hello world'])
    assert abs(m['wer'] - 2 / 6) < 1e-12 and abs(m['cer'] - 5 / 25) < 1e-12
    assert get_cer_wer_metrics(None, None, {}, ['x'], ['']) == {}           # empty reference: jiwer raises, the reference logs and goes on
    m = get_cer_wer_metrics(None, None, {}, ['same text'], ['same text'])
    assert m == {'wer': 0.0, 'cer': 0.0}


def test_eval_rvlcdip_counting_rule_with_scripted_decoder():
    """cruller_eval_rvlcdip.step: 5 greedy steps, a sample scores once if at ANY generated </s> the accumulated text is its
    `<label/>` (ref task_cruller_eval_rvlcdip.py:262-311, incl. the quirk that an early </s> does not end the sample)"""
    from pixparse_amd.framework import DeviceEnv
    from pixparse_amd.models import ImageEncoderCfg, ModelCfg, TextDecoderCfg
    from pixparse_amd.models.archs import register_arch
    from pixparse_amd.task import TaskCrullerEvalRVLCDIP, TaskFactory
    register_arch('vit', 'vit_ft_test', dict(patch=8, dim=128, depth=1, heads=2, mlp_ratio=4, ln_eps=1e-6, pre_norm=False, mean=(0.5,) * 3, std=(0.25,) * 3))
    register_arch('bart', 'bart_ft_test', dict(d_model=128, heads=2, ffn=256, ln_eps=1e-5, vocab=509, dropout=0.0))
    model = ModelCfg(image_encoder=ImageEncoderCfg(name='vit_ft_test', image_fmt='L', image_size=(32, 40), pretrained=False),
                     text_decoder=TextDecoderCfg(name='bart_ft_test', pretrained=False, num_decoder_layers=1, max_length=16))
    task, _ = TaskFactory.create_task('cruller_eval_rvlcdip', dict(model=model, dtype='bfloat16'), DeviceEnv(init_device_type='cpu'), None)
    assert isinstance(task, TaskCrullerEvalRVLCDIP) and task.vocab_size == 50267 + 19 == task.model.vocab_size
    ids = task.tokenizer.trunk.convert_tokens_to_ids
    V = task.vocab_size
    script = [  # per sample: the 5 tokens the decoder "generates"
        [ids('<letter/>'), 2, 5, 5, 5],            # label 0 = letter: correct at the first </s>
        [2, ids('<form/>'), 2, 5, 5],              # label 1 = form: an empty first </s>, right at the second -> counts (quirk)
        [ids('<memo/>'), 2, 5, 5, 5],              # label 2 = email: wrong class
        [ids('<email/>'), 9, 2, 5, 5],             # label 2 = email: extra text before </s> -> no
        [ids('<budget/>'), 2, ids('<budget/>'), 2, 5],   # label 10 = budget: counted once only
    ]

    class Fake:
        def __init__(self):
            self.t = 0
        def image_encoder(self, x):
            return torch.zeros(x.shape[0], 3, 128)
        def decode_begin(self, enc, max_len):
            self.t = 0
        def decode_step(self, ids_in):
            out = torch.full((len(script), V), -1.0)
            for i, s in enumerate(script):
                out[i, s[self.t]] = 1.0
            self.t += 1
            return out
    task.model = Fake()
    sample = {'image': torch.zeros(5, 1, 32, 40), 'label': torch.tensor([0, 1, 2, 2, 10])}
    m = task.step(sample)
    assert m == {'classification': {'correct_samples': 3, 'n_valid_samples': 5}}
    assert task.average_metrics({0: m, 1: m}) == {'classification': {'accuracy': 0.6}}
    import numpy as np
    b = task.collate_fn([{'image': np.zeros((50, 60, 3), np.uint8), 'label': 3}, None, {'image': np.zeros((50, 60), np.uint8), 'label': 4}])
    assert b['image'].shape == (2, 1, 32, 40) and b['label'].tolist() == [3, 4]


def test_eval_metrics_known_answers_and_tree_edit_distance():
    """utils/metrics.py restates Levenshtein / nltk edit_distance / zss (none installable): ANLS and nTED / F1 known answers computed
    by hand, the zss README example, and Zhang-Shasha against an independent exhaustive recursion on random ordered trees with the
    evaluator's own (leaf-aware, non-uniform) costs"""
    import random
    from functools import lru_cache
    from pixparse_amd.utils.metrics import (JSONParseEvaluator, Node, average_normalized_levenshtein_similarity, edit_distance,
                                            normalized_levenshtein, similarity_score, tree_edit_distance)
    assert edit_distance('kitten', 'sitting') == 3 and edit_distance('', 'abc') == 3 and edit_distance('abc', 'abc') == 0
    assert normalized_levenshtein('hello', 'help') == 2 / 5 and similarity_score('hello', 'help') == 1 - 2 / 5
    assert similarity_score('abcd', 'wxyz') == 0                               # nl = 1 >= tau
    assert average_normalized_levenshtein_similarity([['abc', 'abd'], ['hello']], ['abc', 'help']) == (1.0 + 0.6) / 2
    unit = dict(insert_cost=lambda n: 1, remove_cost=lambda n: 1, update_cost=lambda a, b: int(a.label != b.label))
    A = Node('f').addkid(Node('a').addkid(Node('h')).addkid(Node('c').addkid(Node('l')))).addkid(Node('e'))
    B = Node('f').addkid(Node('a').addkid(Node('d')).addkid(Node('c').addkid(Node('b')))).addkid(Node('e'))
    assert tree_edit_distance(A, B, **unit) == 2                               # the example of the zss README
    ev = JSONParseEvaluator()
    gt = {'menu': [{'nm': 'cake', 'cnt': '2'}, {'nm': 'juice', 'cnt': '1'}], 'total': {'price': '9'}}
    pr = {'menu': [{'nm': 'cake', 'cnt': '2'}, {'nm': 'juic', 'cnt': '1'}]}
    assert ev.cal_acc(gt, gt) == 1.0 and ev.cal_f1([gt], [gt]) == 1.0
    # gt tree costs 22 to insert (15 inner nodes + leaves 'cake' '2' 'juice' '1' '9'); pr lacks total->subtree->price->'9' (4) and one letter
    assert abs(ev.cal_acc(pr, gt) - (1 - 5 / 22)) < 1e-12 and abs(ev.cal_f1([pr], [gt]) - 3 / 4.5) < 1e-12
    assert ev.cal_acc({}, gt) == 0 and ev.flatten(ev.normalize_dict(gt))[0] == ('menu.nm', 'cake')
    c = dict(insert_cost=ev.insert_and_remove_cost, remove_cost=ev.insert_and_remove_cost, update_cost=ev.update_cost)

    def freeze(n):
        return (n.label, tuple(freeze(k) for k in n.children))

    @lru_cache(maxsize=None)
    def forest(F, G):      # exhaustive forest edit distance on the rightmost roots (Zhang-Shasha's recurrence without its bookkeeping)
        if not F and not G:
            return 0
        if not G:
            (l, kids) = F[-1]
            return forest(F[:-1] + kids, G) + c['remove_cost'](Node(l))
        if not F:
            (l, kids) = G[-1]
            return forest(F, G[:-1] + kids) + c['insert_cost'](Node(l))
        (lf, kf), (lg, kg) = F[-1], G[-1]
        return min(forest(F[:-1] + kf, G) + c['remove_cost'](Node(lf)), forest(F, G[:-1] + kg) + c['insert_cost'](Node(lg)),
                   forest(F[:-1], G[:-1]) + forest(kf, kg) + c['update_cost'](Node(lf), Node(lg)))

    rng = random.Random(3)

    def rand_tree(depth):
        if depth == 0 or rng.random() < 0.3:
            return Node('<leaf>' + rng.choice(['a', 'ab', 'xyz', 'abcd', '']))
        n = Node(rng.choice(['menu', 'nm', '<subtree>', 'total']))
        for _ in range(rng.randint(1, 3)):
            n.addkid(rand_tree(depth - 1))
        return n
    for _ in range(40):
        a, b = rand_tree(3), rand_tree(3)
        assert tree_edit_distance(a, b, **c) == forest((freeze(a),), (freeze(b),))


def test_eval_docvqa_and_cord_with_scripted_decoder():
    """cruller_eval_docvqa / cruller_eval_cord (ref task_cruller_eval_docvqa.py:270-313, task_cruller_eval_cord.py:335-385): prompt strings,
    string-carried generation with the '</s>' stop rule, token2json of the result, ANLS and nTED / F1 -- the decoder scripted so that
    the expected numbers can be written down; plus generate_string's rule that the decoder is always fed the ids the re-tokenised
    STRING has (the reference's loop), re-prefilling when they are not `previous ids + new id`"""
    from pixparse_amd.framework import DeviceEnv
    from pixparse_amd.models import ImageEncoderCfg, ModelCfg, TextDecoderCfg
    from pixparse_amd.models.archs import register_arch
    from pixparse_amd.task import TaskCrullerEvalCORD, TaskCrullerEvalDOCVQA, TaskFactory
    from pixparse_amd.task.task_cruller_eval_docvqa import generate_string
    import numpy as np
    register_arch('vit', 'vit_ft_test', dict(patch=8, dim=128, depth=1, heads=2, mlp_ratio=4, ln_eps=1e-6, pre_norm=False, mean=(0.5,) * 3, std=(0.25,) * 3))
    register_arch('bart', 'bart_ft_test', dict(d_model=128, heads=2, ffn=256, ln_eps=1e-5, vocab=509, dropout=0.0))
    model = ModelCfg(image_encoder=ImageEncoderCfg(name='vit_ft_test', image_fmt='L', image_size=(32, 40), pretrained=False),
                     text_decoder=TextDecoderCfg(name='bart_ft_test', pretrained=False, num_decoder_layers=1, max_length=128))
    env = DeviceEnv(init_device_type='cpu')

    class Fake:
        """decode path that emits, per sample, the tokens of a scripted answer string; records what it was fed"""
        max_length = 128

        def __init__(self, tok, answers):
            self.tok, self.answers, self.k, self.fed, self.prefills = tok, answers, -1, [], []
        def image_encoder(self, x):
            return torch.zeros(x.shape[0], 3, 128)
        def decode_begin(self, enc, max_len):
            self.cache = []
        def decode_prefill(self, ids):
            self.cache = ids[0].tolist()
            self.prefills.append(list(self.cache))
        def start_sample(self):
            self.k += 1
            self.out = self.tok.encode(self.answers[self.k], add_special_tokens=False)
            self.pos = 0
        def decode_step(self, ids_in):
            self.cache.append(int(ids_in[0, 0]))
            self.fed.append(list(self.cache))
            logits = torch.full((1, len(self.tok)), -1.0)
            logits[0, self.out[self.pos]] = 1.0
            self.pos += 1
            return logits

    # ---- DocVQA
    task, _ = TaskFactory.create_task('cruller_eval_docvqa', dict(model=model, dtype='bfloat16'), env, None)
    assert isinstance(task, TaskCrullerEvalDOCVQA) and task.vocab_size == 50267 + 5 == task.model.vocab_size
    tok = task.tokenizer.trunk
    answers = ['42 USD</s_answer></s>', 'no closing tag</s>']
    fake = Fake(tok, answers)
    task.model = fake
    task.all_ground_truths, task.all_predictions, task.acc_list = [], [], []
    items = [{'image': np.zeros((50, 60), np.uint8), 'labels': {'question': 'total?', 'answers': ['42 USD', '42']}, 'image_id': 1, 'question_id': 10},
             {'image': np.zeros((50, 60, 3), np.uint8), 'labels': {'question': 'who', 'answers': ['x']}, 'image_id': 2, 'question_id': 11}]
    batch = task.collate_fn(items)
    assert batch['images'].shape == (2, 1, 32, 40) and batch['questions'] == ['total?', 'who'] and batch['question_ids'] == [10, 11]
    real_begin = fake.decode_begin
    def counted_begin(enc, max_len):
        real_begin(enc, max_len)
        fake.start_sample()
    fake.decode_begin = counted_begin
    task.step(batch)
    assert task.all_predictions == ['42 USD', ''] and task.all_ground_truths == [['42 USD', '42'], ['x']]
    assert task.average_metrics({}) == {'ANLS': 0.5}
    prompt0 = tok.encode('<s_docvqa><s_question>total?</s_question><s_answer>', add_special_tokens=False)
    assert fake.prefills[0] == prompt0[:-1] and fake.fed[0] == prompt0                      # prefill all but the last prompt token, then feed it
    assert fake.fed[1] == prompt0 + tok.encode('4', add_special_tokens=False)                # every step sees prompt + everything generated
    assert len(fake.prefills) == 2                                                           # one prefill per sample: no re-tokenisation mismatch

    # ---- CORD
    task, _ = TaskFactory.create_task('cruller_eval_cord', dict(model=model, dtype='bfloat16'), env, None)
    assert isinstance(task, TaskCrullerEvalCORD) and task.vocab_size == 50267 + 1 + 54 == task.model.vocab_size
    tok = task.tokenizer.trunk
    gts = [{'gt_parse': {'menu': [{'nm': 'cake', 'cnt': '2'}, {'nm': 'juice', 'cnt': '1'}], 'total': {'total_price': '9'}}},
           {'gt_parse': {'menu': {'nm': 'tea'}}}]
    batch = task.collate_fn([{'image': np.zeros((50, 60), np.uint8), 'ground_truth': repr(g)} for g in gts])
    assert batch['label'].shape == (2, 511) and batch['text_target'].shape == (2, 511) and batch['label'][0, 0] == tok.convert_tokens_to_ids('<s_cord>')
    assert (batch['text_target'][0] == -100).sum() > 400 and batch['text_target'][0, 0] == tok.convert_tokens_to_ids('<s_menu>')
    fake = Fake(tok, ['<s_menu><s_nm>cake</s_nm><s_cnt>2</s_cnt><sep/><s_nm>juic</s_nm><s_cnt>1</s_cnt></s_menu></s>',
                      '<s_menu><s_nm>tea</s_nm></s_menu></s>'])
    real_begin2 = fake.decode_begin
    def counted_begin2(enc, max_len):
        real_begin2(enc, max_len)
        fake.start_sample()
    fake.decode_begin = counted_begin2
    task.model = fake
    task.all_ground_truths, task.all_predictions, task.acc_list = [], [], []
    from pixparse_amd.utils.metrics import JSONParseEvaluator
    task.evaluator = JSONParseEvaluator()
    m = task.step(batch)
    assert m['batch_accuracy'] == 1.0                                                         # the last sample of the batch (ref :383)
    avg = task.average_metrics({0: m})
    assert abs(avg['average_accuracy'] - ((1 - 5 / 22) + 1.0) / 2) < 1e-12 and abs(avg['f1_score'] - 4 / 5.5) < 1e-12
    assert task.all_predictions == [] and fake.prefills == []                                 # one-token prompt: no prefill; state cleared like the reference

    # ---- generate_string follows the STRING: a tokenizer that merges 'a' + 'b' into one id forces a cache rebuild
    class MergeTok:
        eos_token, pad_token_id = '</s>', 1
        vocab = {'<p>': 10, 'a': 11, 'b': 12, 'ab': 13, '</s>': 2}
        def __len__(self):
            return 20
        def encode(self, text, add_special_tokens=False):
            out, i = [], 0
            keys = sorted(self.vocab, key=len, reverse=True)
            while i < len(text):
                for k in keys:
                    if text.startswith(k, i):
                        out.append(self.vocab[k]); i += len(k); break
            return out
        def decode(self, ids):
            inv = {v: k for k, v in self.vocab.items()}
            return ''.join(inv[i] for i in ids)
    class Tk:
        trunk = MergeTok()
    class Fake2(Fake):
        def __init__(self):
            self.fed, self.prefills, self.script, self.pos = [], [], [11, 12, 2], 0
        def decode_step(self, ids_in):
            self.cache.append(int(ids_in[0, 0]))
            self.fed.append(list(self.cache))
            logits = torch.full((1, 20), -1.0)
            logits[0, self.script[self.pos]] = 1.0
            self.pos += 1
            return logits
    f2 = Fake2()
    stats = {}
    text = generate_string(f2, Tk, torch.zeros(3, 8), '<p>', torch.device('cpu'), stats=stats)
    assert text == '<p>ab</s>'
    # after 'a' then 'b' the string '<p>ab' tokenises to [10, 13], not [10, 11, 12]: the decoder is re-prefilled with [10] and fed 13
    assert f2.fed == [[10], [10, 11], [10, 13]] and f2.prefills == [[10]] and stats['prefills'] == 2


# ------------------------------------------------------------------------------------------- pretrained weights / tokenizer
def _tiny_archs():
    from pixparse_amd.models.archs import register_arch
    register_arch('vit', 'vit_pt_test', dict(patch=8, dim=64, depth=2, heads=1, mlp_ratio=2, ln_eps=1e-6, pre_norm=False, mean=(0.5,) * 3, std=(0.5,) * 3))
    register_arch('bart', 'org/bart_pt_test', dict(d_model=64, heads=1, ffn=128, ln_eps=1e-5, vocab=300, dropout=0.0))


def _tiny_cfg(img, fmt, layers, L, enc_pre, dec_pre, path=None):
    from pixparse_amd.models import ImageEncoderCfg, ModelCfg, TextDecoderCfg
    return ModelCfg(image_encoder=ImageEncoderCfg(name='vit_pt_test', image_fmt=fmt, image_size=img, pretrained=enc_pre, pretrained_path=path),
                    text_decoder=TextDecoderCfg(name='org/bart_pt_test', pretrained=dec_pre, num_decoder_layers=layers, max_length=L,
                                                pretrained_path=path))


def test_pretrained_true_without_weights_is_loud(monkeypatch):
    """ref image_encoder_timm.py:13-20 / text_decoder_hf.py:25-31 download weights; offline that must never be silent"""
    from pixparse_amd.models import Cruller
    from pixparse_amd.models.pretrained import PretrainedWeightsMissing
    _tiny_archs()
    monkeypatch.delenv('PIXPARSE_AMD_WEIGHTS', raising=False)
    with pytest.warns(PretrainedWeightsMissing) as rec:
        m = Cruller(_tiny_cfg((16, 24), 'RGB', 2, 16, True, True), vocab_size=300)
    assert len(rec) == 2 and m.pretrained_sources == {'image_encoder': None, 'text_decoder': None}
    monkeypatch.setenv('PIXPARSE_AMD_STRICT_PRETRAINED', '1')
    with pytest.raises(FileNotFoundError, match='RANDOM INITIALISATION'):
        Cruller(_tiny_cfg((16, 24), 'RGB', 2, 16, True, False), vocab_size=300)
    monkeypatch.delenv('PIXPARSE_AMD_STRICT_PRETRAINED')
    import warnings as w
    with w.catch_warnings():
        w.simplefilter('error')
        Cruller(_tiny_cfg((16, 24), 'RGB', 2, 16, False, False), vocab_size=300)      # pretrained=False: nothing to say


def test_pretrained_weights_load_like_timm_and_hf(tmp_path, monkeypatch):
    """a synthetic timm ViT checkpoint (3-channel 32x32 -> 4x4 grid) into a 1-channel 16x40 model (adapt_input_conv +
    resample_abs_pos_embed), and a synthetic full seq2seq BART checkpoint (6 decoder layers, `model.shared.weight`) into a
    2-layer causal decoder: first n layers, tied + resizable embeddings"""
    import torch.nn.functional as F
    from safetensors.torch import save_file
    from pixparse_amd.layers.engines import BartEngine, ViTEngine
    from pixparse_amd.models import Cruller
    from pixparse_amd.models.archs import BART_ARCHS, VIT_ARCHS
    _tiny_archs()
    g = torch.Generator().manual_seed(0)
    rnd = lambda *s: torch.randn(*s, generator=g)
    va, ba = VIT_ARCHS['vit_pt_test'], BART_ARCHS['org/bart_pt_test']
    vit_sd = {k: rnd(*shape) for k, shape in ViTEngine.param_shapes(va, 3, (32, 32))}
    vit_sd['head.weight'] = rnd(10, 64)                                   # classifier head of the checkpoint: ignored (num_classes=0)
    V, L_ck = 300, 32
    bart_sd = {'model.shared.weight': rnd(V, 64), 'model.encoder.layers.0.fc1.weight': rnd(128, 64), 'final_logits_bias': torch.zeros(1, V)}
    for item in BartEngine.param_shapes(ba, 6, V, L_ck):
        if not item[0].endswith('embed_tokens.weight'):
            bart_sd[item[0]] = rnd(*item[1])
    save_file(vit_sd, str(tmp_path / 'vit_pt_test.safetensors'))
    save_file(bart_sd, str(tmp_path / 'org--bart_pt_test.safetensors'))
    monkeypatch.setenv('PIXPARSE_AMD_WEIGHTS', str(tmp_path))
    import warnings as w
    with w.catch_warnings():
        w.simplefilter('error')                                           # found -> no warning
        m = Cruller(_tiny_cfg((16, 40), 'L', 2, L_ck, True, True))
    assert m.vocab_size == V and all(p and p.startswith(str(tmp_path)) for p in m.pretrained_sources.values())
    sd = m.state_dict()
    e, d = 'image_encoder.trunk.', 'text_decoder.trunk.'
    assert torch.equal(sd[e + 'blocks.1.mlp.fc2.weight'], vit_sd['blocks.1.mlp.fc2.weight'])
    assert torch.allclose(sd[e + 'patch_embed.proj.weight'], vit_sd['patch_embed.proj.weight'].sum(1, keepdim=True))
    pos = vit_sd['pos_embed']
    grid = F.interpolate(pos[:, 1:].reshape(1, 4, 4, 64).permute(0, 3, 1, 2), size=(2, 5), mode='bicubic', antialias=True, align_corners=False)
    want = torch.cat([pos[:, :1], grid.permute(0, 2, 3, 1).reshape(1, 10, 64)], 1)
    assert sd[e + 'pos_embed'].shape == (1, 11, 64) and torch.allclose(sd[e + 'pos_embed'], want, atol=1e-6)
    for i in range(2):                                                    # the FIRST n decoder layers of the checkpoint
        k = f'model.decoder.layers.{i}.encoder_attn.k_proj.weight'
        assert torch.equal(sd[d + k], bart_sd[k])
    assert d + 'model.decoder.layers.2.fc1.weight' not in sd
    assert torch.equal(sd[d + 'model.decoder.embed_tokens.weight'], bart_sd['model.shared.weight'])
    assert sd[d + 'lm_head.weight'].data_ptr() == sd[d + 'model.decoder.embed_tokens.weight'].data_ptr()
    assert torch.equal(sd[d + 'model.decoder.embed_positions.weight'], bart_sd['model.decoder.embed_positions.weight'])
    m.text_decoder.trunk.resize_token_embeddings(V + 2)                   # what the task does after adding 2 tokens (Q7)
    sd2 = m.state_dict()
    assert sd2[d + 'lm_head.weight'].shape == (V + 2, 64) and torch.equal(sd2[d + 'lm_head.weight'][:V], bart_sd['model.shared.weight'])
    assert torch.equal(sd2[e + 'blocks.0.attn.qkv.weight'], vit_sd['blocks.0.attn.qkv.weight'])


def test_tokenizer_fallback_is_explicit_or_loud(monkeypatch):
    from pixparse_amd.tokenizers import BYTE_TOKENIZER, ByteBartTokenizer, TokenizerCfg, TokenizerFallbackWarning, TokenizerHF
    import warnings as w
    with w.catch_warnings():
        w.simplefilter('error')
        t = TokenizerHF(TokenizerCfg(name=BYTE_TOKENIZER))                # explicit opt-in: silent
    assert isinstance(t.trunk, ByteBartTokenizer) and len(t.trunk) == 50265
    with pytest.warns(TokenizerFallbackWarning, match='NOT compatible'):
        t = TokenizerHF(TokenizerCfg(name='no-such-org/no-such-tokenizer'))
    assert isinstance(t.trunk, ByteBartTokenizer)
    monkeypatch.setenv('PIXPARSE_AMD_STRICT_TOKENIZER', '1')
    with pytest.raises(Exception):
        TokenizerHF(TokenizerCfg(name='no-such-org/no-such-tokenizer'))


def test_head_dim_is_checked():
    """a head_dim other than 64 must not run silently (kernels index head h at channel 64 h)"""
    from pixparse_amd import hip, ops
    from pixparse_amd.layers.arena import ParamArena
    from pixparse_amd.layers.engines import BartEngine, Buffers, ViTEngine
    q = torch.zeros(1, 8, 96)
    with pytest.raises(ValueError, match='head_dim 64'):
        ops.attn_fwd(q, q, q, q, torch.zeros(1, 3, 8), 3, 0.1, False)
    with pytest.raises(ValueError, match='head_dim 64'):
        ViTEngine(dict(patch=8, dim=96, depth=1, heads=3, mlp_ratio=4, ln_eps=1e-6, pre_norm=False), 3, (16, 16), ParamArena(), '', Buffers('cpu'))
    with pytest.raises(ValueError, match='head_dim 64'):
        BartEngine(dict(d_model=128, heads=4, ffn=256, ln_eps=1e-5), 1, 100, 16, ParamArena(), '', Buffers('cpu'))
    hip.load()
    with pytest.raises(hip.HipLibraryError, match='head_dim is 64'):   # row stride 64 cannot hold 2 heads of 64 channels
        hip.call('crl_attn_fwd', 16, 512, 64, 16, 512, 64, 16, 512, 64, 16, 512, 64, 16, 1, 2, 8, 8, 0.125, 0, 0, 0.0, 0, 0, 0, None)


# ------------------------------------------------------------------------------------------- DeviceEnv under torchrun, one rank
def test_device_env_single_torchrun_rank_takes_the_distributed_branch():
    """WORLD_SIZE=1 exported by torchrun still builds a process group (gloo on a CPU env) and an ACTIVE reducer, so
    `bench.py --gpus 1` under the launcher walks the code `--gpus 8` does; without the launcher env nothing is created"""
    import socket
    import subprocess
    import sys
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    code = '''
import torch, torch.distributed as dist
from pixparse_amd.framework import DeviceEnv
from pixparse_amd.framework.reducer import BucketedGradReducer
from pixparse_amd.layers.arena import ParamArena
env = DeviceEnv('cpu')
assert env.distributed and env.world_size == 1 and dist.is_initialized() and dist.get_backend() == 'gloo'
a = ParamArena(); a.add('w', (1000,)); a.add('b', (10,)); a.materialize('cpu'); a.alloc_training_state()
r = BucketedGradReducer(a, env.world_size, bucket_bytes=1024, active=env.distributed)
assert r.active and len(r.buckets) > 2
a.g.fill_(3.0); r.begin(); r.on_ready('b'); r.on_ready('w'); r.finish()
assert float(a.g.min()) == 3.0 and r.grad_divisor() == 1.0 and r._next == len(r.buckets)
r.broadcast_params(0)
dist.destroy_process_group()
print('OK')
'''
    env = dict(os.environ, RANK='0', LOCAL_RANK='0', WORLD_SIZE='1', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, cwd=ROOT, env=env, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith('OK'), r.stderr[-2000:]
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT', 'LOCAL_RANK')}
    code2 = "from pixparse_amd.framework import DeviceEnv; e = DeviceEnv('cpu'); assert not e.distributed and e.world_size == 1; print('OK')"
    r = subprocess.run([sys.executable, '-c', code2], capture_output=True, text=True, cwd=ROOT, env=env, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith('OK'), r.stderr[-2000:]


def test_finetune_xent_is_registered_and_refuses_swin():
    from pixparse_amd.models import Cruller, ImageEncoderCfg, ModelCfg, TextDecoderCfg
    from pixparse_amd.task import TaskCrullerFinetuneXent, TaskFactory
    assert TaskFactory.TASK_CLASS_REGISTRY['cruller_finetune_xent'][0] is TaskCrullerFinetuneXent
    m = Cruller(ModelCfg(image_encoder=ImageEncoderCfg(name='swin_tiny_patch4_window7_224', image_size=(224, 224), pretrained=False),
                         text_decoder=TextDecoderCfg(name='facebook/bart-base', pretrained=False, num_decoder_layers=1, max_length=8)), vocab_size=300)
    with pytest.raises(NotImplementedError, match='GetCLSToken'):
        m.add_classifier_head(16)


def test_stale_library_is_refused(monkeypatch):
    """hip.load() compares the sha1 of every kernel source with what build.py recorded when it built the library: a library older than
    its sources must fail loudly (it would run last week's kernels), with an explicit escape for the hand-built A/B scripts"""
    from pixparse_amd import build, hip
    if not os.path.exists(build.BUILD_INFO):
        pytest.skip('library was not built through pixparse_amd.build')
    hip._check_not_stale(hip.LIB_PATH)                                   # as built: fine
    real = build.source_digest()
    monkeypatch.setattr(build, 'source_digest', lambda: {**real, 'attention.hip': '0' * 40})
    with pytest.raises(hip.HipLibraryError, match='attention.hip'):
        hip._check_not_stale(hip.LIB_PATH)


def test_gemm_wave_quantisation_cut_decisions():
    """the cost model of the wave-quantisation cut (gemm.hip quant_rows, measured in round 3) seen through crl_gemm_ws_bytes: a cut shows
    as the split-K scratch of its 360 remainder rows (49 512 = 192 x 256 + 360).  Out-width 1024 with a contraction >= 2048 is cut after
    three whole rounds (8 / 8 / 4 slabs of the remainder); K = 1024 never is; the GELU / dGELU remainders cannot be slabbed and stay uncut."""
    from pixparse_amd import hip, ops
    hip.load()
    M, rem = 49512, 360
    slab = rem * 1024 * 4
    q = lambda *a: hip.query('crl_gemm_ws_bytes', *a)
    assert q(hip.NN, ops.EPI_BF16, M, 1024, 4096) == 8 * slab          # fc1 dgrad
    assert q(hip.NN, ops.EPI_BF16, M, 1024, 3072) == 6 * slab          # qkv dgrad (48 K tiles: 6 slabs of 8)
    assert q(hip.NN, ops.EPI_BF16, M, 1024, 2048) == 4 * slab          # decoder k/v dgrad
    assert q(hip.NT, ops.EPI_F32_RESID, M, 1024, 4096) == 8 * slab     # fc2 + residual
    for layout, epi, N, K in ((hip.NN, ops.EPI_BF16, 1024, 1024), (hip.NT, ops.EPI_BF16, 3072, 1024), (hip.NT, ops.EPI_F32_RESID, 1024, 1024),
                              (hip.NT, ops.EPI_BF16_GELU, 4096, 1024), (hip.NN, ops.EPI_BF16_DGELU, 4096, 1024)):
        assert q(layout, epi, M, N, K) == 0, (layout, epi, N, K)
    hip.call('crl_gemm_set_quant_cost', -1.0)                           # the tuning knob switches the cut off altogether
    try:
        assert q(hip.NN, ops.EPI_BF16, M, 1024, 4096) == 0
    finally:
        hip.call('crl_gemm_set_quant_cost', 1.0)


def test_every_knob_is_documented_and_tested():
    """every process-wide entry point (crl_*_set_*) declared in include/crl.h and every PIXPARSE_AMD_* variable the package
    reads has a line in INTEGRATION.md, and every set_ entry point is flipped by at least one test; nothing documented there is gone from the code"""
    import glob
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    header = open(os.path.join(root, 'include', 'crl.h')).read()
    doc = open(os.path.join(root, 'INTEGRATION.md')).read()
    tests = ''.join(open(f).read() for f in glob.glob(os.path.join(root, 'tests', '*.py')) if not f.endswith('test_host_cpu.py')) + \
        open(os.path.join(root, 'tests', 'test_host_cpu.py')).read().split('def test_every_knob_is_documented_and_tested')[0]
    setters = sorted(set(re.findall(r'^int (crl_\w+_set_\w+)\(', header, re.M)))
    assert len(setters) >= 12, setters
    for fn in setters:
        assert f'`{fn}' in doc, f'{fn} has no line in INTEGRATION.md'
        assert f"'{fn}'" in tests or f'{fn}(' in tests, f'{fn} is not flipped by any test'
    pkg = ''
    for dirpath, _, files in os.walk(os.path.join(root, 'pixparse_amd')):
        for f in files:
            if f.endswith('.py'):
                pkg += open(os.path.join(dirpath, f)).read()
    envs = sorted(set(re.findall(r'PIXPARSE_AMD_[A-Z0-9_]+', pkg)))
    for e in envs:
        assert e in doc, f'{e} is read by the package but not documented in INTEGRATION.md'
    removed = doc.split('Removed in round 6')[1] if 'Removed in round 6' in doc else ''
    for e in sorted(set(re.findall(r'PIXPARSE_AMD_[A-Z0-9_]+', doc))):
        if e in removed and e not in doc.split('Removed in round 6')[0]:
            assert e not in pkg, f'{e} is documented as removed but still read'
        elif e != 'PIXPARSE_AMD_LIVE_ORACLE':                      # (a switch of the test suite, not of the package)
            assert e in pkg or e in tests, f'{e} is documented but nothing reads it'
    for fn in re.findall(r'`(crl_\w+_set_\w+)', removed):
        assert fn not in header, f'{fn} is documented as removed but still declared'
