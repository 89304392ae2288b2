"""Explicit forward / backward engines of the Cruller model over the HIP C-ABI.

No autograd, no tracing: every layer's forward stores exactly the activations its hand-written
backward needs in named device buffers, and the backward sweep runs the dgrad / wgrad / LayerNorm /
attention kernels in reverse order, accumulating weight gradients straight into the flat fp32
gradient arena.  Dtype flow = CUDA autocast(bf16) policy: the residual stream
and LayerNorm statistics are fp32, every GEMM / attention operand and result is bf16 with fp32
accumulation.

Arithmetic restated (reference call sites):
  ViT / CLIP-ViT  timm VisionTransformer built at  models/image_encoder_timm.py:13-20
  Swin            timm SwinTransformer (same call site; output flattened to [B, HW, C])
  BART decoder    transformers BartForCausalLM built at  models/text_decoder_hf.py:13-33
  loss            nn.CrossEntropyLoss(ignore_index=-100)  task/task_cruller_pretrain.py:118,251-256
"""
from typing import Callable, Dict, Optional, Tuple

import os

import torch

from .. import ops
from ..ops import BF16, F16, F32, EPI_BF16, EPI_BF16_GELU, EPI_BF16_DGELU, EPI_F32_RESID, EPI_F32, EPI_F32_ACC
from .arena import ParamArena


_FUSE_CROSS_KV = os.environ.get('PIXPARSE_AMD_FUSE_CROSS_KV', '1') != '0'           # A/B switch: one K/V projection GEMM for all decoder layers
_DECODE_LN_FUSION = os.environ.get('PIXPARSE_AMD_DECODE_LN_FUSION', '1') != '0'     # A/B switch of the generation step (scripts/bench_generate.py)


_POISON = os.environ.get('PIXPARSE_AMD_POISON', '0') == '1'


class Buffers:
    """named device buffers, allocated on first use and reused every step (static activation plan)."""

    def __init__(self, device):
        self.device = device
        self.t: Dict[str, torch.Tensor] = {}

    def get(self, name: str, shape, dtype, zero: bool = False) -> torch.Tensor:
        shape = tuple(int(s) for s in shape)
        t = self.t.get(name)
        if t is None or t.shape != shape or t.dtype != dtype:
            with torch.inference_mode(False):   # a buffer first sized inside an eval task's inference_mode() block must stay writable outside it
                t = (torch.zeros if zero else torch.empty)(shape, dtype=dtype, device=self.device)
                if _POISON and not zero and t.numel():      # debug aid: a buffer that is read before it is written shows up as NaN (0xFF.. in every dtype)
                    t.reshape(-1).view(torch.uint8).fill_(255)
            self.t[name] = t
        return t

    def bytes(self) -> int:
        return sum(t.numel() * t.element_size() for t in self.t.values())


class _Base:
    def __init__(self, arena: ParamArena, prefix: str, bufs: Buffers, tag: str):
        self.arena, self.prefix, self.bufs, self.tag = arena, prefix, bufs, tag

    def P(self, n):
        return self.arena.param(self.prefix + n)

    def G(self, n):
        return self.arena.grad(self.prefix + n)

    def W(self, n):
        return self.arena.shadow(self.prefix + n)

    def buf(self, name, shape, dtype, zero=False):
        return self.bufs.get(self.tag + '.' + name, shape, dtype, zero)

    def has(self, n):
        return (self.prefix + n) in self.arena.entries

    # LN helpers -----------------------------------------------------------------
    def ln_fwd(self, name, x, key, eps, want_f32=False, want_bf16=True):
        M, D = x.shape
        mean = self.buf(key + '.mean', (M,), F32)
        rstd = self.buf(key + '.rstd', (M,), F32)
        y32 = self.buf(key + '.y32', (M, D), F32) if want_f32 else None
        y16 = self.buf(key + '.y16', (M, D), BF16) if want_bf16 else None
        ops.layernorm_fwd(x, self.P(name + '.weight'), self.P(name + '.bias'), eps, y32, y16, mean, rstd)
        return y32, y16

    def ln_bwd(self, name, key, x, dy_f32, dy_bf16, dx_f32, acc, dx_bf16, bias_of=None):
        """bias_of: name of the Linear whose OUTPUT gradient dx_bf16 is -- its bias gradient (column sums of dx_bf16) is
        accumulated by the same kernel instead of a separate colsum pass (lin_wgrad is then called with has_bias=False)"""
        ops.layernorm_bwd(dy_f32, dy_bf16, x, self.P(name + '.weight'), self.bufs.t[self.tag + '.' + key + '.mean'],
                          self.bufs.t[self.tag + '.' + key + '.rstd'], dx_f32, acc, dx_bf16,
                          self.G(name + '.weight'), self.G(name + '.bias'), True,
                          dx_colsum=self.G(bias_of + '.bias') if bias_of else None)

    # Linear backward: wgrad + bias grad (dgrad is issued by the caller: its epilogue differs)
    # round 6: on the FIRST micro-step of an accumulation window (every step at grad_accum_steps = 1) the weight-gradient GEMMs and the bias
    # gradients that ride them OVERWRITE the gradient arena instead of adding to the zeros AdamW left there: the split-K reduce / the fp32
    # epilogue skip the read of 2.1 GB of old values per step (same bits: 0 + s == s).  Set by CrullerModel.backward(first_micro=...); every
    # other producer (LayerNorm, embeddings, position tables: sparse or small) keeps accumulating into the zero-filled arena.
    first_micro = False

    @property
    def acc(self) -> bool:
        return not self.first_micro

    def lin_wgrad(self, name, dy, x, has_bias=True, n=None, k=None):
        gw = self.G(name + '.weight')
        gw2 = gw.view(gw.shape[0], -1)
        # the bias gradient (column sums of dy) rides the weight-gradient GEMM, which streams dy anyway
        ops.linear_wgrad(dy, x, gw2, self.acc, n=n, k=k, dbias=self.G(name + '.bias') if has_bias else None, dbias_accumulate=self.acc)


# ============================================================================================ ViT
class ViTEngine(_Base):
    def __init__(self, arch: dict, in_chans: int, img_size: Tuple[int, int], arena, prefix, bufs):
        super().__init__(arena, prefix, bufs, 'vit')
        self.a = arch
        self.C = in_chans
        self.H, self.Wd = img_size
        self.P_ = arch['patch']
        self.gh, self.gw = self.H // self.P_, self.Wd // self.P_
        self.Np = self.gh * self.gw
        self.N = self.Np + 1
        self.D = arch['dim']
        self.heads = arch['heads']
        if self.D != self.heads * ops.HEAD_DIM:
            raise ValueError(f"ViT arch dim {self.D} / heads {self.heads}: the attention kernels are built for head_dim {ops.HEAD_DIM}")
        self.F = self.D * arch['mlp_ratio']
        self.Kreal = in_chans * self.P_ * self.P_
        self.Kp = ops.round_up(self.Kreal, ops.K_PAD)
        self.pe_shadow = None  # [D, Kp] bf16 padded copy of the patch-embed weight

    @staticmethod
    def param_shapes(arch, in_chans, img_size):
        P, D = arch['patch'], arch['dim']
        gh, gw = img_size[0] // P, img_size[1] // P
        F_ = D * arch['mlp_ratio']
        s = [('cls_token', (1, 1, D)), ('pos_embed', (1, gh * gw + 1, D)), ('patch_embed.proj.weight', (D, in_chans, P, P))]
        if not arch['pre_norm']:
            s.append(('patch_embed.proj.bias', (D,)))
        else:
            s += [('norm_pre.weight', (D,)), ('norm_pre.bias', (D,))]
        for i in range(arch['depth']):
            bp = f'blocks.{i}.'
            s += [(bp + 'norm1.weight', (D,)), (bp + 'norm1.bias', (D,)),
                  (bp + 'attn.qkv.weight', (3 * D, D)), (bp + 'attn.qkv.bias', (3 * D,)),
                  (bp + 'attn.proj.weight', (D, D)), (bp + 'attn.proj.bias', (D,)),
                  (bp + 'norm2.weight', (D,)), (bp + 'norm2.bias', (D,)),
                  (bp + 'mlp.fc1.weight', (F_, D)), (bp + 'mlp.fc1.bias', (F_,)),
                  (bp + 'mlp.fc2.weight', (D, F_)), (bp + 'mlp.fc2.bias', (D,))]
        s += [('norm.weight', (D,)), ('norm.bias', (D,))]
        return s

    def out_tokens(self):
        return self.N

    def refresh_shadows(self):
        """padded bf16 copy of the conv weight as a [D, Kp] GEMM operand (K = C*P*P is not a multiple of 32)."""
        if self.pe_shadow is None:
            self.pe_shadow = torch.zeros(self.D, self.Kp, dtype=BF16, device=self.bufs.device)
        w = self.P('patch_embed.proj.weight')
        ops.cast_pad_bf16(w, self.pe_shadow, self.D, self.Kreal, self.Kp)

    def forward(self, image: torch.Tensor):
        a, B, D, N, F_ = self.a, image.shape[0], self.D, self.N, self.F
        M = B * N
        self.B = B
        eps = a['ln_eps']
        patches = self.buf('patches', (B * self.Np, self.Kp), BF16)
        ops.im2row(image, patches, self.P_, self.gh, self.gw)
        pe = self.buf('pe', (B * self.Np, D), BF16)
        ops.linear_fwd(patches, self.pe_shadow, None if a['pre_norm'] else self.P('patch_embed.proj.bias'), pe)
        x = self.buf('x0', (M, D), F32)
        ops.vit_tokens_fwd(pe, self.P('cls_token'), self.P('pos_embed'), x, B, self.Np, D)
        if a['pre_norm']:
            x, _ = self.ln_fwd('norm_pre', x, 'norm_pre', eps, want_f32=True, want_bf16=False)
        scale = (D // self.heads) ** -0.5
        for i in range(a['depth']):
            bp, k = f'blocks.{i}.', f'b{i}'
            _, h1 = self.ln_fwd(bp + 'norm1', x, k + '.ln1', eps)
            qkv = self.buf(k + '.qkv', (M, 3 * D), BF16)
            # the q columns leave the GEMM as q * scale * log2(e) (one rounding): the flash kernels get base-2 logits from the MFMAs
            ops.linear_fwd(h1, self.W(bp + 'attn.qkv.weight'), self.P(bp + 'attn.qkv.bias'), qkv, colscale=scale * ops.LOG2E, colscale_cols=D)
            q3 = qkv.view(B, N, 3 * D)
            o = self.buf(k + '.o', (M, D), BF16)
            lse = self.buf(k + '.lse', (B, self.heads, N), F32)
            ops.attn_fwd(q3[:, :, 0:D], q3[:, :, D:2 * D], q3[:, :, 2 * D:3 * D], o.view(B, N, D), lse, self.heads, scale, False, q_prescaled=True)
            x2 = self.buf(k + '.x2', (M, D), F32)
            ops.linear_fwd(o, self.W(bp + 'attn.proj.weight'), self.P(bp + 'attn.proj.bias'), x2, EPI_F32_RESID, resid=x)
            _, h2 = self.ln_fwd(bp + 'norm2', x2, k + '.ln2', eps)
            pre = self.buf(k + '.dact', (M, F_), F16)     # gelu'(fc1 output), saved by the GELU epilogue for the fc2 dgrad epilogue
            act = self.buf(k + '.act', (M, F_), BF16)
            ops.linear_fwd(h2, self.W(bp + 'mlp.fc1.weight'), self.P(bp + 'mlp.fc1.bias'), act, EPI_BF16_GELU, aux=pre)
            x3 = self.buf(k + '.x3', (M, D), F32)
            ops.linear_fwd(act, self.W(bp + 'mlp.fc2.weight'), self.P(bp + 'mlp.fc2.bias'), x3, EPI_F32_RESID, resid=x2)
            self._save_in(k, x)
            x = x3
        self.x_last = x
        enc32, enc16 = self.ln_fwd('norm', x, 'norm', eps, want_f32=True, want_bf16=True)
        return enc32, enc16

    def _save_in(self, k, x):
        self.bufs.t[self.tag + '.' + k + '.xin'] = x  # alias: block input is the previous block's x3 / x0 / norm_pre out

    def backward(self, denc: torch.Tensor, on_layer_done: Optional[Callable[[str], None]] = None):
        """denc: fp32 [B*N, D] gradient of the encoder output."""
        a, B, D, N, F_ = self.a, self.B, self.D, self.N, self.F
        M = B * N
        T = self.bufs.t
        tg = self.tag + '.'
        dx = self.buf('dx', (M, D), F32)
        gb = self.buf('gb', (M, D), BF16)
        last = a['depth'] - 1
        self.ln_bwd('norm', 'norm', self.x_last, denc, None, dx, False, gb, bias_of=f'blocks.{last}.mlp.fc2')
        if on_layer_done:
            on_layer_done(self.prefix + 'norm.weight')
        scale = (D // self.heads) ** -0.5
        dpre = self.buf('dpre', (M, F_), BF16)
        dh = self.buf('dh', (M, D), BF16)
        do = self.buf('do', (M, D), BF16)
        dqkv = self.buf('dqkv', (M, 3 * D), BF16)
        delta = self.buf('delta', (2, B, self.heads, N), F32)
        for i in reversed(range(a['depth'])):
            bp, k = f'blocks.{i}.', f'b{i}'
            xin, x2 = T[tg + k + '.xin'], T[tg + k + '.x2']
            h1, h2 = T[tg + k + '.ln1.y16'], T[tg + k + '.ln2.y16']
            qkv, o, lse = T[tg + k + '.qkv'], T[tg + k + '.o'], T[tg + k + '.lse']
            pre, act = T[tg + k + '.dact'], T[tg + k + '.act']
            # ---- MLP: x3 = x2 + fc2(gelu(fc1(LN2(x2))));  gb = bf16(dx3)
            # (the bias gradients of fc2 / attn.proj = column sums of gb were accumulated by the LayerNorm backward that wrote gb)
            ops.linear_dgrad(gb, self.W(bp + 'mlp.fc2.weight'), dpre, EPI_BF16_DGELU, aux=pre)
            self.lin_wgrad(bp + 'mlp.fc2', gb, act, has_bias=False)
            ops.linear_dgrad(dpre, self.W(bp + 'mlp.fc1.weight'), dh)
            self.lin_wgrad(bp + 'mlp.fc1', dpre, h2)
            self.ln_bwd(bp + 'norm2', k + '.ln2', x2, None, dh, dx, True, gb, bias_of=bp + 'attn.proj')     # dx := dx2, gb := bf16(dx2)
            # ---- attention: x2 = xin + proj(attn(qkv(LN1(xin))))
            ops.linear_dgrad(gb, self.W(bp + 'attn.proj.weight'), do)
            self.lin_wgrad(bp + 'attn.proj', gb, o, has_bias=False)
            q3, dq3 = qkv.view(B, N, 3 * D), dqkv.view(B, N, 3 * D)
            ops.attn_bwd(q3[:, :, 0:D], q3[:, :, D:2 * D], q3[:, :, 2 * D:], o.view(B, N, D), do.view(B, N, D), lse, delta,
                         dq3[:, :, 0:D], dq3[:, :, D:2 * D], dq3[:, :, 2 * D:], self.heads, scale, False, q_prescaled=True)
            ops.linear_dgrad(dqkv, self.W(bp + 'attn.qkv.weight'), dh)
            self.lin_wgrad(bp + 'attn.qkv', dqkv, h1)
            self.ln_bwd(bp + 'norm1', k + '.ln1', xin, None, dh, dx, True, gb,    # dx := d(xin) = d(x3 of block i-1)
                        bias_of=f'blocks.{i - 1}.mlp.fc2' if i > 0 else None)
            if on_layer_done:
                on_layer_done(self.prefix + bp + 'norm1.weight')
        if a['pre_norm']:
            self.ln_bwd('norm_pre', 'norm_pre', T[tg + 'x0'], dx, None, dx, False, None)
        dpe = T[tg + 'pe']
        ops.vit_tokens_bwd(dx, dpe, self.G('cls_token'), self.G('pos_embed'), B, self.Np, D, True)
        self.lin_wgrad('patch_embed.proj', dpe, T[tg + 'patches'], has_bias=not a['pre_norm'], k=self.Kreal)
        if on_layer_done:
            on_layer_done(self.prefix + 'cls_token')


# ============================================================================================ Swin
class SwinEngine(_Base):
    def __init__(self, arch: dict, in_chans: int, img_size: Tuple[int, int], arena, prefix, bufs):
        super().__init__(arena, prefix, bufs, 'swin')
        self.a, self.C_in = arch, in_chans
        self.H, self.Wd = img_size
        self.geo = self.stage_geometry(arch, img_size)
        self.P_ = arch['patch']
        self.Kreal = in_chans * self.P_ * self.P_
        self.Kp = ops.round_up(self.Kreal, ops.K_PAD)
        self.pe_shadow = None
        # drop-path (timm DropPath on both residual branches; rate of block j of n = drop.p_path * j / (n - 1)) of the NEXT forward /
        # backward pair: an ops.DropSpec with p_path > 0, or None.  Sites 2 j (attention branch) and 2 j + 1 (MLP branch).
        self.drop: Optional[ops.DropSpec] = None
        self.nblocks = sum(arch['depths'])

    def _path_rate(self, j: int) -> float:
        if self.drop is None or not self.drop.p_path or self.nblocks < 2:
            return 0.0
        return self.drop.p_path * j / (self.nblocks - 1)

    def _join(self, x, w, b, out, resid, key, j, branch, rows):
        """out(f32) = resid + [drop_path](bf16(x @ w^T + b)); the per-sample scales stay in `key` for the backward pass"""
        rate = self._path_rate(j)
        if rate <= 0.0:
            ops.linear_fwd(x, w, b, out, EPI_F32_RESID, resid=resid)
            return
        tmp = self.buf('dp.tmp', (x.shape[0], w.shape[0]), BF16)
        ops.linear_fwd(x, w, b, tmp)
        scale = self.buf(key, (self.B,), F32)
        ops.droppath_scale(scale, rate, self.drop, 2 * j + branch)
        ops.rowscale_add(tmp, scale, resid, out, rows)

    def _branch_grad(self, gb, key, j, rows, name):
        """gradient entering a (possibly dropped) branch: gb itself, or bf16(gb * scale[sample]) in a scratch buffer"""
        if self._path_rate(j) <= 0.0:
            return gb
        out = self.buf(name, gb.shape, BF16)
        ops.rowscale_bf16(gb, self.bufs.t[self.tag + '.' + key], out, rows)
        return out

    @staticmethod
    def stage_geometry(arch, img_size):
        Hf, Wf = img_size[0] // arch['patch'], img_size[1] // arch['patch']
        out = []
        for si, depth in enumerate(arch['depths']):
            if si > 0:
                Hf, Wf = Hf // 2, Wf // 2
            C = arch['embed_dim'] * (2 ** si)
            w = arch['window']
            if min(Hf, Wf) <= w:
                w = min(Hf, Wf)
            out.append((Hf, Wf, C, arch['heads'][si], w, depth))
        return out

    @staticmethod
    def param_shapes(arch, in_chans, img_size):
        P, C0 = arch['patch'], arch['embed_dim']
        s = [('patch_embed.proj.weight', (C0, in_chans, P, P)), ('patch_embed.proj.bias', (C0,)),
             ('patch_embed.norm.weight', (C0,)), ('patch_embed.norm.bias', (C0,))]
        for si, (Hf, Wf, C, heads, w, depth) in enumerate(SwinEngine.stage_geometry(arch, img_size)):
            sp = f'layers.{si}.'
            if si > 0:
                s += [(sp + 'downsample.norm.weight', (2 * C,)), (sp + 'downsample.norm.bias', (2 * C,)),
                      (sp + 'downsample.reduction.weight', (C, 2 * C))]
            F_ = C * arch['mlp_ratio']
            for bi in range(depth):
                bp = sp + f'blocks.{bi}.'
                s += [(bp + 'norm1.weight', (C,)), (bp + 'norm1.bias', (C,)),
                      (bp + 'attn.qkv.weight', (3 * C, C)), (bp + 'attn.qkv.bias', (3 * C,)),
                      (bp + 'attn.relative_position_bias_table', ((2 * w - 1) ** 2, heads)),
                      (bp + 'attn.proj.weight', (C, C)), (bp + 'attn.proj.bias', (C,)),
                      (bp + 'norm2.weight', (C,)), (bp + 'norm2.bias', (C,)),
                      (bp + 'mlp.fc1.weight', (F_, C)), (bp + 'mlp.fc1.bias', (F_,)),
                      (bp + 'mlp.fc2.weight', (C, F_)), (bp + 'mlp.fc2.bias', (C,))]
        Cl = C0 * 2 ** (len(arch['depths']) - 1)
        s += [('norm.weight', (Cl,)), ('norm.bias', (Cl,))]
        return s

    def out_tokens(self):
        Hf, Wf = self.geo[-1][0], self.geo[-1][1]
        return Hf * Wf

    def refresh_shadows(self):
        C0 = self.a['embed_dim']
        if self.pe_shadow is None:
            self.pe_shadow = torch.zeros(C0, self.Kp, dtype=BF16, device=self.bufs.device)
        ops.cast_pad_bf16(self.P('patch_embed.proj.weight'), self.pe_shadow, C0, self.Kreal, self.Kp)

    def forward(self, image):
        a, B = self.a, image.shape[0]
        self.B = B
        eps = a['ln_eps']
        Hf0, Wf0, C0 = self.geo[0][0], self.geo[0][1], self.geo[0][2]
        patches = self.buf('patches', (B * Hf0 * Wf0, self.Kp), BF16)
        ops.im2row(image, patches, self.P_, Hf0, Wf0)
        pe = self.buf('pe', (B * Hf0 * Wf0, C0), BF16)
        ops.linear_fwd(patches, self.pe_shadow, self.P('patch_embed.proj.bias'), pe)
        pe32 = self.buf('pe32', (B * Hf0 * Wf0, C0), F32)
        ops.add_bf16_to_f32(pe, pe32, False)
        x, _ = self.ln_fwd('patch_embed.norm', pe32, 'pe_norm', eps, want_f32=True, want_bf16=False)
        gj = 0          # global block index (drop-path schedule)
        for si, (Hf, Wf, C, heads, w, depth) in enumerate(self.geo):
            sp, M = f'layers.{si}.', B * Hf * Wf
            if si > 0:
                Cp = C // 2
                mg = self.buf(f's{si}.merged', (M, 4 * Cp), F32)
                ops.patch_merge_fwd(x, mg, B, Hf * 2, Wf * 2, Cp)
                _, mh = self.ln_fwd(sp + 'downsample.norm', mg, f's{si}.dsn', eps)
                red = self.buf(f's{si}.red', (M, C), BF16)
                ops.linear_fwd(mh, self.W(sp + 'downsample.reduction.weight'), None, red)
                x = self.buf(f's{si}.x', (M, C), F32)
                ops.add_bf16_to_f32(red, x, False)
            F_ = C * a['mlp_ratio']
            scale = (C // heads) ** -0.5
            for bi in range(depth):
                bp, k = sp + f'blocks.{bi}.', f's{si}b{bi}'
                shift = 0 if (bi % 2 == 0 or w >= min(Hf, Wf)) else w // 2
                _, h1 = self.ln_fwd(bp + 'norm1', x, k + '.ln1', eps)
                qkv = self.buf(k + '.qkv', (M, 3 * C), BF16)
                ops.linear_fwd(h1, self.W(bp + 'attn.qkv.weight'), self.P(bp + 'attn.qkv.bias'), qkv)
                o = self.buf(k + '.o', (M, C), BF16)
                ops.swin_attn_fwd(qkv, self.P(bp + 'attn.relative_position_bias_table'), o, B, Hf, Wf, heads, w, shift, scale)
                x2 = self.buf(k + '.x2', (M, C), F32)
                self._join(o, self.W(bp + 'attn.proj.weight'), self.P(bp + 'attn.proj.bias'), x2, x, k + '.dps1', gj, 0, Hf * Wf)
                _, h2 = self.ln_fwd(bp + 'norm2', x2, k + '.ln2', eps)
                pre = self.buf(k + '.dact', (M, F_), F16)     # gelu'(fc1 output), saved by the GELU epilogue for the fc2 dgrad epilogue
                act = self.buf(k + '.act', (M, F_), BF16)
                ops.linear_fwd(h2, self.W(bp + 'mlp.fc1.weight'), self.P(bp + 'mlp.fc1.bias'), act, EPI_BF16_GELU, aux=pre)
                x3 = self.buf(k + '.x3', (M, C), F32)
                self._join(act, self.W(bp + 'mlp.fc2.weight'), self.P(bp + 'mlp.fc2.bias'), x3, x2, k + '.dps2', gj, 1, Hf * Wf)
                self.bufs.t[self.tag + '.' + k + '.xin'] = x
                x = x3
                gj += 1
        self.x_last = x
        return self.ln_fwd('norm', x, 'norm', eps, want_f32=True, want_bf16=True)

    def backward(self, denc, on_layer_done=None):
        a, B, T, tg = self.a, self.B, self.bufs.t, self.tag + '.'
        Hl, Wl, Cl = self.geo[-1][0], self.geo[-1][1], self.geo[-1][2]
        dx = self.buf(f'dx{len(self.geo) - 1}', (B * Hl * Wl, Cl), F32)
        gb = self.buf(f'gb{len(self.geo) - 1}', (B * Hl * Wl, Cl), BF16)
        self.ln_bwd('norm', 'norm', self.x_last, denc, None, dx, False, gb)
        if on_layer_done:
            on_layer_done(self.prefix + 'norm.weight')
        gj_end = self.nblocks
        for si in reversed(range(len(self.geo))):
            Hf, Wf, C, heads, w, depth = self.geo[si]
            gj_end -= depth          # global index of this stage's first block
            sp, M = f'layers.{si}.', B * Hf * Wf
            F_ = C * a['mlp_ratio']
            scale = (C // heads) ** -0.5
            dpre = self.buf(f'dpre{si}', (M, F_), BF16)
            dh = self.buf(f'dh{si}', (M, C), BF16)
            do = self.buf(f'do{si}', (M, C), BF16)
            dqkv = self.buf(f'dqkv{si}', (M, 3 * C), BF16)
            for bi in reversed(range(depth)):
                bp, k = sp + f'blocks.{bi}.', f's{si}b{bi}'
                shift = 0 if (bi % 2 == 0 or w >= min(Hf, Wf)) else w // 2
                xin, x2 = T[tg + k + '.xin'], T[tg + k + '.x2']
                h1, h2 = T[tg + k + '.ln1.y16'], T[tg + k + '.ln2.y16']
                qkv, o, pre, act = T[tg + k + '.qkv'], T[tg + k + '.o'], T[tg + k + '.dact'], T[tg + k + '.act']
                gj = gj_end + bi
                gm = self._branch_grad(gb, k + '.dps2', gj, Hf * Wf, f'gbs{si}')      # d(MLP branch output) = drop-path scale o d(x3)
                ops.linear_dgrad(gm, self.W(bp + 'mlp.fc2.weight'), dpre, EPI_BF16_DGELU, aux=pre)
                self.lin_wgrad(bp + 'mlp.fc2', gm, act)
                ops.linear_dgrad(dpre, self.W(bp + 'mlp.fc1.weight'), dh)
                self.lin_wgrad(bp + 'mlp.fc1', dpre, h2)
                self.ln_bwd(bp + 'norm2', k + '.ln2', x2, None, dh, dx, True, gb)
                ga = self._branch_grad(gb, k + '.dps1', gj, Hf * Wf, f'gbs{si}')      # d(attention branch output)
                ops.linear_dgrad(ga, self.W(bp + 'attn.proj.weight'), do)
                self.lin_wgrad(bp + 'attn.proj', ga, o)
                ops.swin_attn_bwd(qkv, self.P(bp + 'attn.relative_position_bias_table'), do, dqkv,
                                  self.G(bp + 'attn.relative_position_bias_table'), B, Hf, Wf, heads, w, shift, scale)
                ops.linear_dgrad(dqkv, self.W(bp + 'attn.qkv.weight'), dh)
                self.lin_wgrad(bp + 'attn.qkv', dqkv, h1)
                self.ln_bwd(bp + 'norm1', k + '.ln1', xin, None, dh, dx, True, gb)
                if on_layer_done:
                    on_layer_done(self.prefix + bp + 'norm1.weight')
            if si > 0:
                # x_stage = float(reduction(LN(merge(x_prev))))  -> gb is bf16(dx) = d(reduction out)
                Cp = C // 2
                mh, mg = T[tg + f's{si}.dsn.y16'], T[tg + f's{si}.merged']
                dmh = self.buf(f'dmh{si}', (M, 4 * Cp), BF16)
                ops.linear_dgrad(gb, self.W(sp + 'downsample.reduction.weight'), dmh)
                self.lin_wgrad(sp + 'downsample.reduction', gb, mh, has_bias=False)
                dmg = self.buf(f'dmg{si}', (M, 4 * Cp), F32)
                self.ln_bwd(sp + 'downsample.norm', f's{si}.dsn', mg, None, dmh, dmg, False, None)
                dx = self.buf(f'dx{si - 1}', (B * Hf * 2 * Wf * 2, Cp), F32)
                gb = self.buf(f'gb{si - 1}', (B * Hf * 2 * Wf * 2, Cp), BF16)
                ops.patch_merge_bwd(dmg, dx, B, Hf * 2, Wf * 2, Cp)
                ops.cast_bf16(dx, gb)
                if on_layer_done:
                    on_layer_done(self.prefix + sp + 'downsample.norm.weight')
        # patch embed: x = LN(float(conv(img)))
        Hf0, Wf0, C0 = self.geo[0][0], self.geo[0][1], self.geo[0][2]
        dpe32 = self.buf('dpe32', (B * Hf0 * Wf0, C0), F32)
        dpe = self.buf('dpe', (B * Hf0 * Wf0, C0), BF16)
        self.ln_bwd('patch_embed.norm', 'pe_norm', T[tg + 'pe32'], dx, None, dpe32, False, dpe)
        self.lin_wgrad('patch_embed.proj', dpe, T[tg + 'patches'], has_bias=True, k=self.Kreal)
        if on_layer_done:
            on_layer_done(self.prefix + 'patch_embed.proj.weight')


# ============================================================================================ BART decoder
class BartEngine(_Base):
    DP = 'model.decoder.'

    def __init__(self, arch: dict, n_layers: int, vocab: int, max_pos: int, arena, prefix, bufs):
        super().__init__(arena, prefix, bufs, 'dec')
        self.a, self.L, self.V, self.max_pos = arch, n_layers, vocab, max_pos
        self.Vp = ops.round_up(vocab, ops.VOCAB_PAD)
        self.D, self.heads, self.F = arch['d_model'], arch['heads'], arch['ffn']
        if self.D != self.heads * ops.HEAD_DIM:
            raise ValueError(f"BART arch d_model {self.D} / heads {self.heads}: the attention kernels are built for head_dim {ops.HEAD_DIM}")
        # hidden-state dropout of the NEXT forward / backward pair (ops.DropSpec) or None; set by Cruller when training dropout
        # is switched on.  Sites: 0 = embedding LayerNorm output, 1 + 3 i + {0, 1, 2} = self-attention /
        # cross-attention / fc2 branch of layer i before its residual join (modeling_bart.py:362,377,384-386,654).
        self.drop: Optional[ops.DropSpec] = None

    def _hidden_drop(self):
        return self.drop is not None and self.drop.p > 0

    def _branch(self, x, w, b, out, resid, site):
        """out(f32) = resid + [dropout](bf16(x @ w^T + b)): the residual join behind an attention / FFN branch"""
        if not self._hidden_drop():
            ops.linear_fwd(x, w, b, out, EPI_F32_RESID, resid=resid)
        else:
            tmp = self.buf('drop.tmp', (x.shape[0], w.shape[0]), BF16)
            ops.linear_fwd(x, w, b, tmp)
            ops.dropout_add(tmp, resid, out, self.drop, site)

    def _branch_bwd(self, ln_name, key, x, dy32, dyb, dx32, dxb, bias_of, site):
        """LayerNorm backward behind a residual join; dxb = the (dropped) gradient of the branch's Linear output, whose bias
        gradient is its column sum (fused into the LayerNorm backward when there is no mask to apply first)"""
        if not self._hidden_drop():
            self.ln_bwd(ln_name, key, x, dy32, dyb, dx32, False, dxb, bias_of=bias_of)
        else:
            self.ln_bwd(ln_name, key, x, dy32, dyb, dx32, False, dxb)
            ops.dropout(dxb, dxb, self.drop, site)
            ops.colsum(dxb, self.G(bias_of + '.bias'), True)

    @staticmethod
    def param_shapes(arch, n_layers, vocab, max_pos):
        """arena order: embeddings first, then per layer [self q,k,v] [out] LN [cross q] [cross k,v] [out] LN fc1 fc2 LN
        (q/k/v and k/v stay adjacent so one fused GEMM covers them)."""
        D, F_ = arch['d_model'], arch['ffn']
        assert D % 64 == 0, 'd_model must be a multiple of 64 (fused q/k/v arena views, head_dim 64)'
        dp = BartEngine.DP
        Vp = ops.round_up(vocab, ops.VOCAB_PAD)
        s = [(dp + 'embed_tokens.weight', (vocab, D), Vp * D), (dp + 'embed_positions.weight', (max_pos + 2, D)),
             (dp + 'layernorm_embedding.weight', (D,)), (dp + 'layernorm_embedding.bias', (D,))]
        for i in range(n_layers):
            lp = dp + f'layers.{i}.'
            for a in ('self_attn', 'encoder_attn'):
                for proj in ('q_proj', 'k_proj', 'v_proj'):
                    s.append((lp + f'{a}.{proj}.weight', (D, D)))
                for proj in ('q_proj', 'k_proj', 'v_proj'):
                    s.append((lp + f'{a}.{proj}.bias', (D,)))
                s += [(lp + f'{a}.out_proj.weight', (D, D)), (lp + f'{a}.out_proj.bias', (D,)),
                      (lp + f'{a}_layer_norm.weight', (D,)), (lp + f'{a}_layer_norm.bias', (D,))]
            s += [(lp + 'fc1.weight', (F_, D)), (lp + 'fc1.bias', (F_,)), (lp + 'fc2.weight', (D, F_)), (lp + 'fc2.bias', (D,)),
                  (lp + 'final_layer_norm.weight', (D,)), (lp + 'final_layer_norm.bias', (D,))]
        return s

    # fused views: q,k,v (or k,v) weights/biases are adjacent ALIGN-padded entries of equal size ------------------
    def _fused(self, kind, first, count, rows):
        e = self.arena.entries[self.prefix + first]
        flat = {'p': self.arena.p, 'g': self.arena.g, 'w': self.arena.pb}[kind]
        return flat[e.offset:e.offset + count * e.numel].view(*rows)

    def fw(self, kind, lp, attn, first, count):   # fused weight [count*D, D]
        return self._fused(kind, lp + f'{attn}.{first}.weight', count, (count * self.D, self.D))

    def fb(self, kind, lp, attn, first, count):   # fused bias [count*D]
        return self._fused(kind, lp + f'{attn}.{first}.bias', count, (count * self.D,))

    def _kv_fused(self, Me: int) -> bool:
        return self.L > 1 and (Me + 256) * self.L * 2 * self.D * 2 < (1 << 32) and _FUSE_CROSS_KV

    def _kv_all(self, enc16, Me):
        if not self._kv_fused(Me):
            return None
        dp, D, L = self.DP, self.D, self.L
        w_all = self.buf('kv.w_all', (L * 2 * D, D), BF16)
        b_all = self.buf('kv.b_all', (L * 2 * D,), F32)
        for i in range(L):
            lp = dp + f'layers.{i}.'
            w_all[2 * D * i:2 * D * (i + 1)].copy_(self.fw('w', lp, 'encoder_attn', 'k_proj', 2))
            b_all[2 * D * i:2 * D * (i + 1)].copy_(self.fb('p', lp, 'encoder_attn', 'k_proj', 2))
        kv_all = self.buf('kv.all', (Me, L * 2 * D), BF16)
        ops.linear_fwd(enc16, w_all, b_all, kv_all)
        return kv_all

    def forward(self, ids: torch.Tensor, enc16: torch.Tensor, S: int):
        """ids [B, T] int64; enc16 bf16 [B*S, D] -> logits buffer bf16 [B*T, Vp]."""
        dp, D, F_, H = self.DP, self.D, self.F, self.heads
        B, T = ids.shape
        self.B, self.T, self.S, self.ids = B, T, S, ids
        M, Me = B * T, B * S
        eps = self.a['ln_eps']
        scale = (D // H) ** -0.5
        emb = self.buf('emb', (M, D), F32)
        ops.embed_fwd(ids, self.P(dp + 'embed_tokens.weight'), self.P(dp + 'embed_positions.weight'), emb, 2)
        h, hb = self.ln_fwd(dp + 'layernorm_embedding', emb, 'ln_emb', eps, want_f32=True, want_bf16=True)
        if self._hidden_drop():
            ops.dropout(h, h, self.drop, 0, y_bf16=hb)
        # attention-probability dropout (sites 200 + 2 i: self, 201 + 2 i: cross; hf BartAttention dropout = config.attention_dropout) and
        # activation dropout behind the GELU (site 300 + i; hf:384): live only when the DropSpec carries p_attn / p_act (bart-base)
        drop, p_act = self.drop, (self.drop.p_act if self.drop is not None else 0.0)
        # The cross-attention K / V projections of ALL layers read the same encoder output: one GEMM [Me, D] x [L * 2D, D]^T into one
        # [Me, L * 2D] buffer (layer i's k | v are column block i), and in backward() one dgrad GEMM over the L * 2D columns instead of L
        # GEMMs that each read-modify-write the fp32 d(encoder output) (cfg-3: 10 x 267 us -> 1.66 ms).  The weights of the L layers are
        # gathered into one operand per step (L copies of 4 MB).  Off when the fused buffer would pass the GEMM's 4 GiB operand limit.
        kv_all = self._kv_all(enc16, Me)
        for i in range(self.L):
            lp, k = dp + f'layers.{i}.', f'l{i}'
            qkv = self.buf(k + '.qkv', (M, 3 * D), BF16)
            ops.linear_fwd(hb, self.fw('w', lp, 'self_attn', 'q_proj', 3), self.fb('p', lp, 'self_attn', 'q_proj', 3), qkv,
                           colscale=scale * ops.LOG2E, colscale_cols=D)          # q leaves the projection as q * scale * log2(e)
            q3 = qkv.view(B, T, 3 * D)
            o1 = self.buf(k + '.o1', (M, D), BF16)
            lse1 = self.buf(k + '.lse1', (B, H, T), F32)
            ops.attn_fwd(q3[:, :, 0:D], q3[:, :, D:2 * D], q3[:, :, 2 * D:], o1.view(B, T, D), lse1, H, scale, True, drop=drop, site=200 + 2 * i,
                         q_prescaled=True)
            t1 = self.buf(k + '.t1', (M, D), F32)
            self._branch(o1, self.W(lp + 'self_attn.out_proj.weight'), self.P(lp + 'self_attn.out_proj.bias'), t1, h, 1 + 3 * i)
            h1, h1b = self.ln_fwd(lp + 'self_attn_layer_norm', t1, k + '.ln1', eps, True, True)
            q2 = self.buf(k + '.q2', (M, D), BF16)
            ops.linear_fwd(h1b, self.W(lp + 'encoder_attn.q_proj.weight'), self.P(lp + 'encoder_attn.q_proj.bias'), q2,
                           colscale=scale * ops.LOG2E, colscale_cols=D)
            if kv_all is not None:
                kv3 = kv_all.view(B, S, self.L * 2 * D)[:, :, 2 * D * i:2 * D * (i + 1)]
            else:
                kv2 = self.buf(k + '.kv2', (Me, 2 * D), BF16)
                ops.linear_fwd(enc16, self.fw('w', lp, 'encoder_attn', 'k_proj', 2), self.fb('p', lp, 'encoder_attn', 'k_proj', 2), kv2)
                kv3 = kv2.view(B, S, 2 * D)
            o2 = self.buf(k + '.o2', (M, D), BF16)
            lse2 = self.buf(k + '.lse2', (B, H, T), F32)
            ops.attn_fwd(q2.view(B, T, D), kv3[:, :, 0:D], kv3[:, :, D:], o2.view(B, T, D), lse2, H, scale, False, drop=drop, site=201 + 2 * i,
                         q_prescaled=True)
            t2 = self.buf(k + '.t2', (M, D), F32)
            self._branch(o2, self.W(lp + 'encoder_attn.out_proj.weight'), self.P(lp + 'encoder_attn.out_proj.bias'), t2, h1, 2 + 3 * i)
            h2, h2b = self.ln_fwd(lp + 'encoder_attn_layer_norm', t2, k + '.ln2', eps, True, True)
            pre = self.buf(k + '.dact', (M, F_), F16)     # gelu'(fc1 output), saved by the GELU epilogue for the fc2 dgrad epilogue
            act = self.buf(k + '.act', (M, F_), BF16)
            ops.linear_fwd(h2b, self.W(lp + 'fc1.weight'), self.P(lp + 'fc1.bias'), act, EPI_BF16_GELU, aux=pre)
            if p_act:
                ops.dropout(act, act, drop, 300 + i, p=p_act)      # the saved activation IS the dropped one (what fc2 and its wgrad read)
            t3 = self.buf(k + '.t3', (M, D), F32)
            self._branch(act, self.W(lp + 'fc2.weight'), self.P(lp + 'fc2.bias'), t3, h2, 3 + 3 * i)
            self.bufs.t[self.tag + '.' + k + '.hb'] = hb
            h, hb = self.ln_fwd(lp + 'final_layer_norm', t3, k + '.ln3', eps, True, True)
        self.h_last16 = hb
        logits = self.buf('logits', (M, self.Vp), BF16)
        ops.linear_fwd(hb, self.arena.shadow(self.prefix + dp + 'embed_tokens.weight', padded=True).view(self.Vp, D), None, logits)
        return logits

    # ------------------------------------------------------------------ generation with a KV cache
    def _lin(self, x, w, b, out, epi=ops.EPI_BF16, resid=None):
        """decode rows go through the skinny kernel (weights read once), prefill-sized inputs through the GEMM"""
        if x.shape[0] <= ops.SKINNY_MAX_ROWS:
            ops.linear_skinny(x, w, b, out, epi, resid=resid)
        else:
            aux = self.buf('gen.gelu_dact', out.shape, F16) if epi == EPI_BF16_GELU else None   # the GEMM epilogue also stores gelu'(h) (unused here)
            ops.linear_fwd(x, w, b, out, epi, aux=aux, resid=resid)

    def _ln_lin(self, ln_name, t32, key, eps, w, b, out, epi=ops.EPI_BF16, **row):
        """h = LayerNorm(t32), out = epilogue(bf16(h) @ w^T + b); returns the fp32 h (the next residual).  Decode rows: ONE launch
        (crl_linear_skinny_ln_bf16, every workgroup normalises the <= 16 rows for itself); more rows: ln_fwd + the GEMM"""
        if (t32.shape[0] <= ops.SKINNY_MAX_ROWS and w.shape[0] >= t32.shape[1] and _DECODE_LN_FUSION
                and t32.shape[0] * (t32.shape[1] + 32) * 2 <= 60 * 1024):          # the normalised rows live in LDS
            h = self.buf(key + '.y32', tuple(t32.shape), F32)
            ops.linear_skinny_ln(t32, self.P(ln_name + '.weight'), self.P(ln_name + '.bias'), eps, h, w, b, out, epi, **row)
            return h
        h, hb = self.ln_fwd(ln_name, t32, key, eps, True, True)
        if row:
            assert hb.shape[0] <= ops.SKINNY_MAX_ROWS, 'a device-selected output row needs the skinny kernel (<= 16 sequences)'
            ops.linear_skinny(hb, w, b, out, epi, **row)
        else:
            self._lin(hb, w, b, out, epi)
        return h

    def decode_begin(self, enc16: torch.Tensor, B: int, S: int, max_len: int):
        """enc16 bf16 [B*S, D].  Projects the encoder states to every layer's cross-attention K/V once and lays out
        empty self-attention caches [B, max_len, 3D] (q | k | v per row, read by the attention kernel as strided views).
        The step counter lives on the device (`gen.step`): every launch of decode_step has identical arguments, so one
        captured hipGraph replays the whole step."""
        assert max_len <= self.max_pos, f'max_len {max_len} exceeds the {self.max_pos} learned positions'
        dp, D = self.DP, self.D
        step = self.buf('gen.step', (1,), torch.int32)
        step.zero_()
        self.gen = dict(B=B, S=S, max_len=max_len, t=0, step=step)
        for i in range(self.L):
            lp = dp + f'layers.{i}.'
            kv2 = self.buf(f'gen.l{i}.kv2', (B * S, 2 * D), BF16)
            ops.linear_fwd(enc16, self.fw('w', lp, 'encoder_attn', 'k_proj', 2), self.fb('p', lp, 'encoder_attn', 'k_proj', 2), kv2)
            self.buf(f'gen.l{i}.kvc', (B, max_len, 3 * D), BF16)      # q | k | v of every step: one fused projection per token

    def decode_prefill(self, ids: torch.Tensor) -> None:
        """ids [B, P]: the first P tokens of a multi-token prompt (every sequence the same length), positions 0..P-1 of an empty
        cache.  ONE pass of the whole decoder over the P rows -- the training forward's layer arithmetic (fused q|k|v projection,
        causal flash attention, cross-attention over the K/V that decode_begin projected) -- whose q|k|v rows land in cache rows
        0..P-1; no logits are produced (the caller feeds the prompt's LAST token through decode_step, which yields the first
        generated token).  ref: the reference re-runs the decoder on the whole prompt for every generated token
        (task/task_cruller_eval_docvqa.py:279-297, task_cruller_eval_cord.py:345-372)."""
        g = self.gen
        B, S, step = g['B'], g['S'], g['step']
        P_ = ids.shape[1]
        assert ids.shape[0] == B and g['t'] == 0 and 0 < P_ < g['max_len'], 'prefill starts from an empty cache and must leave room to decode'
        dp, D, F_, H = self.DP, self.D, self.F, self.heads
        M = B * P_
        eps = self.a['ln_eps']
        scale = (D // H) ** -0.5
        emb = self.buf('pre.emb', (M, D), F32)
        ops.embed_fwd(ids, self.P(dp + 'embed_tokens.weight'), self.P(dp + 'embed_positions.weight'), emb, 2)
        h, hb = self.ln_fwd(dp + 'layernorm_embedding', emb, 'pre.ln_emb', eps, want_f32=True, want_bf16=True)
        for i in range(self.L):
            lp, k = dp + f'layers.{i}.', f'pre.l{i}'
            kvc = self.bufs.t[self.tag + f'.gen.l{i}.kvc']
            qkv = self.buf('pre.qkv', (M, 3 * D), BF16)
            self._lin(hb, self.fw('w', lp, 'self_attn', 'q_proj', 3), self.fb('p', lp, 'self_attn', 'q_proj', 3), qkv)
            rows = kvc[:, :P_, :]
            rows.copy_(qkv.view(B, P_, 3 * D))                          # cache rows 0..P-1 (memory movement only)
            o1 = self.buf('pre.o1', (M, D), BF16)
            lse = self.buf('pre.lse', (B, H, P_), F32)
            ops.attn_fwd(rows[:, :, 0:D], rows[:, :, D:2 * D], rows[:, :, 2 * D:], o1.view(B, P_, D), lse, H, scale, True)
            t1 = self.buf('pre.t1', (M, D), F32)
            self._lin(o1, self.W(lp + 'self_attn.out_proj.weight'), self.P(lp + 'self_attn.out_proj.bias'), t1, EPI_F32_RESID, resid=h)
            h1, h1b = self.ln_fwd(lp + 'self_attn_layer_norm', t1, 'pre.ln1', eps, True, True)
            q2 = self.buf('pre.q2', (M, D), BF16)
            self._lin(h1b, self.W(lp + 'encoder_attn.q_proj.weight'), self.P(lp + 'encoder_attn.q_proj.bias'), q2)
            kv3 = self.bufs.t[self.tag + f'.gen.l{i}.kv2'].view(B, S, 2 * D)
            o2 = self.buf('pre.o2', (M, D), BF16)
            ops.attn_fwd(q2.view(B, P_, D), kv3[:, :, 0:D], kv3[:, :, D:], o2.view(B, P_, D), lse, H, scale, False)
            t2 = self.buf('pre.t2', (M, D), F32)
            self._lin(o2, self.W(lp + 'encoder_attn.out_proj.weight'), self.P(lp + 'encoder_attn.out_proj.bias'), t2, EPI_F32_RESID, resid=h1)
            h2, h2b = self.ln_fwd(lp + 'encoder_attn_layer_norm', t2, 'pre.ln2', eps, True, True)
            act = self.buf('pre.act', (M, F_), BF16)
            self._lin(h2b, self.W(lp + 'fc1.weight'), self.P(lp + 'fc1.bias'), act, EPI_BF16_GELU)
            t3 = self.buf('pre.t3', (M, D), F32)
            self._lin(act, self.W(lp + 'fc2.weight'), self.P(lp + 'fc2.bias'), t3, EPI_F32_RESID, resid=h2)
            h, hb = self.ln_fwd(lp + 'final_layer_norm', t3, f'pre.ln3.{i & 1}', eps, True, True)   # h / hb of layer i feed layer i + 1: two alternating buffers
        step.fill_(P_)
        g['t'] = P_

    def decode_step(self, ids: torch.Tensor) -> torch.Tensor:
        """ids [B, 1] = the token at position step (= number of tokens fed so far) -> logits bf16 [B, Vp] for the next
        position. Same layer arithmetic as forward(); the self-attention sees the cached keys 0..step, nothing is
        recomputed. Safe to capture in a hipGraph (no host reads, no allocation after the first call)."""
        g = self.gen
        B, S, step = g['B'], g['S'], g['step']
        assert ids.shape == (B, 1) and g['t'] < g['max_len'], 'KV cache is full'
        dp, D, F_, H = self.DP, self.D, self.F, self.heads
        eps = self.a['ln_eps']
        scale = (D // H) ** -0.5
        emb = self.buf('gen.emb', (B, D), F32)
        ops.embed_decode(ids, self.P(dp + 'embed_tokens.weight'), self.P(dp + 'embed_positions.weight'), emb, step, 2)
        # post-LN BART: every LayerNorm feeds exactly one projection (bf16) and one residual add (fp32), so it rides in FRONT of that
        # projection's launch (crl_linear_skinny_ln_bf16) -- 3 launches per layer less on the dependent chain of the step
        t_in, ln_in, ln_key = emb, dp + 'layernorm_embedding', 'gen.ln_emb'
        for i in range(self.L):
            lp, k = dp + f'layers.{i}.', f'gen.l{i}'
            kvc = self.bufs.t[self.tag + '.' + k + '.kvc']
            # q | k | v of this token in ONE projection, straight into cache row `step`; the attention takes q from that row
            h = self._ln_lin(ln_in, t_in, ln_key, eps, self.fw('w', lp, 'self_attn', 'q_proj', 3), self.fb('p', lp, 'self_attn', 'q_proj', 3),
                             kvc[:, 0, :], out_row=step, out_row_stride=3 * D)
            o1 = self.buf(k + '.o1', (B, D), BF16)
            ops.attn_decode(kvc[:, 0, 0:D], kvc[:, :, D:2 * D], kvc[:, :, 2 * D:], o1, H, scale, nk_minus1=step,
                            q_row=step, q_row_stride=3 * D)
            t1 = self.buf(k + '.t1', (B, D), F32)
            self._lin(o1, self.W(lp + 'self_attn.out_proj.weight'), self.P(lp + 'self_attn.out_proj.bias'), t1, EPI_F32_RESID, resid=h)
            q2 = self.buf(k + '.q2', (B, D), BF16)
            h1 = self._ln_lin(lp + 'self_attn_layer_norm', t1, k + '.ln1', eps, self.W(lp + 'encoder_attn.q_proj.weight'),
                              self.P(lp + 'encoder_attn.q_proj.bias'), q2)
            kv3 = self.bufs.t[self.tag + '.' + k + '.kv2'].view(B, S, 2 * D)
            o2 = self.buf(k + '.o2', (B, D), BF16)
            ops.attn_decode(q2, kv3[:, :, 0:D], kv3[:, :, D:], o2, H, scale)
            t2 = self.buf(k + '.t2', (B, D), F32)
            self._lin(o2, self.W(lp + 'encoder_attn.out_proj.weight'), self.P(lp + 'encoder_attn.out_proj.bias'), t2, EPI_F32_RESID, resid=h1)
            act = self.buf(k + '.act', (B, F_), BF16)
            h2 = self._ln_lin(lp + 'encoder_attn_layer_norm', t2, k + '.ln2', eps, self.W(lp + 'fc1.weight'), self.P(lp + 'fc1.bias'), act, EPI_BF16_GELU)
            t3 = self.buf(k + '.t3', (B, D), F32)
            self._lin(act, self.W(lp + 'fc2.weight'), self.P(lp + 'fc2.bias'), t3, EPI_F32_RESID, resid=h2)
            t_in, ln_in, ln_key = t3, lp + 'final_layer_norm', k + '.ln3'
        _, hb = self.ln_fwd(ln_in, t_in, ln_key, eps, False, True)      # the LM head has 3144 workgroups: its LayerNorm stays a launch of its own
        logits = self.buf('gen.logits', (B, self.Vp), BF16)
        self._lin(hb, self.arena.shadow(self.prefix + dp + 'embed_tokens.weight', padded=True).view(self.Vp, D), None, logits)
        step.add_(1)
        g['t'] += 1
        return logits

    def backward(self, dlogits: torch.Tensor, enc16: torch.Tensor, denc: torch.Tensor, on_layer_done=None):
        """dlogits bf16 [M, Vp] (pad columns zero); writes the encoder-output gradient to denc (fp32 [B*S, D]; overwritten, not
        accumulated: no zero fill needed)."""
        dp, D, F_, H = self.DP, self.D, self.F, self.heads
        B, T, S = self.B, self.T, self.S
        M, Me = B * T, B * S
        Tb, tg = self.bufs.t, self.tag + '.'
        scale = (D // H) ** -0.5
        Ew = self.arena.shadow(self.prefix + dp + 'embed_tokens.weight', padded=True).view(self.Vp, D)
        Eg = self.arena.grad(self.prefix + dp + 'embed_tokens.weight', padded=True).view(self.Vp, D)
        dyb = self.buf('dyb', (M, D), BF16)        # bf16 gradient arriving at a layer output from GEMM consumers
        ops.linear_dgrad(dlogits, Ew, dyb)
        ops.linear_wgrad(dlogits, self.h_last16, Eg, self.acc)       # (tied weight: the embedding backward below ADDS its rows to this)
        dy32 = None                                 # fp32 gradient arriving through the residual path
        dt = self.buf('dt', (M, D), F32)
        dtb = self.buf('dtb', (M, D), BF16)
        dt2 = self.buf('dt2', (M, D), F32)
        dpre = self.buf('dpre', (M, F_), BF16)
        dhb = self.buf('dhb', (M, D), BF16)
        do = self.buf('do', (M, D), BF16)
        dq2 = self.buf('dq2', (M, D), BF16)
        fused_kv = self._kv_fused(Me)
        dkv_all = self.buf('kv.dall', (Me, self.L * 2 * D), BF16) if fused_kv else None
        dkv2 = None if fused_kv else self.buf('dkv2', (Me, 2 * D), BF16)
        dqkv = self.buf('dqkv', (M, 3 * D), BF16)
        delta = self.buf('delta', (2, B, H, T), F32)
        if self.L == 0:
            denc.zero_()
        drop, p_act = self.drop, (self.drop.p_act if self.drop is not None else 0.0)
        for i in reversed(range(self.L)):
            lp, k = dp + f'layers.{i}.', f'l{i}'
            g = lambda n: Tb[tg + k + n]
            # ---- h_out = LN3(t3), t3 = h2 + fc2(gelu(fc1(h2b)))
            self._branch_bwd(lp + 'final_layer_norm', k + '.ln3', g('.t3'), dy32, dyb, dt, dtb, lp + 'fc2', 3 + 3 * i)
            ops.linear_dgrad(dtb, self.W(lp + 'fc2.weight'), dpre, EPI_BF16_DGELU, aux=g('.dact'))
            if p_act:
                ops.dropout(dpre, dpre, drop, 300 + i, p=p_act)    # mask o scale and gelu' are both elementwise: applied behind the fused dGELU epilogue
            self.lin_wgrad(lp + 'fc2', dtb, g('.act'), has_bias=False)
            ops.linear_dgrad(dpre, self.W(lp + 'fc1.weight'), dhb)
            self.lin_wgrad(lp + 'fc1', dpre, g('.ln2.y16'))
            # ---- h2 = LN2(t2), t2 = h1 + out_c(attn(q_c(h1b), kv_c(enc)))
            self._branch_bwd(lp + 'encoder_attn_layer_norm', k + '.ln2', g('.t2'), dt, dhb, dt2, dtb, lp + 'encoder_attn.out_proj', 2 + 3 * i)
            ops.linear_dgrad(dtb, self.W(lp + 'encoder_attn.out_proj.weight'), do)
            self.lin_wgrad(lp + 'encoder_attn.out_proj', dtb, g('.o2'), has_bias=False)
            if fused_kv:
                kv3 = Tb[tg + 'kv.all'].view(B, S, self.L * 2 * D)[:, :, 2 * D * i:2 * D * (i + 1)]
                dkv2 = dkv_all[:, 2 * D * i:2 * D * (i + 1)]                  # column block i: a strided [Me, 2D] operand
                dkv3 = dkv_all.view(B, S, self.L * 2 * D)[:, :, 2 * D * i:2 * D * (i + 1)]
            else:
                kv3, dkv3 = g('.kv2').view(B, S, 2 * D), dkv2.view(B, S, 2 * D)
            ops.attn_bwd(g('.q2').view(B, T, D), kv3[:, :, 0:D], kv3[:, :, D:], g('.o2').view(B, T, D), do.view(B, T, D), g('.lse2'), delta,
                         dq2.view(B, T, D), dkv3[:, :, 0:D], dkv3[:, :, D:], H, scale, False, drop=drop, site=201 + 2 * i, q_prescaled=True)
            ops.linear_dgrad(dq2, self.W(lp + 'encoder_attn.q_proj.weight'), dhb)
            self.lin_wgrad(lp + 'encoder_attn.q_proj', dq2, g('.ln1.y16'))
            # d(encoder output) accumulates over the layers: the first one written (the last layer) overwrites, so denc needs no zero fill
            if not fused_kv:
                ops.linear_dgrad(dkv2, self.fw('w', lp, 'encoder_attn', 'k_proj', 2), denc, EPI_F32 if i == self.L - 1 else EPI_F32_ACC)
            ops.linear_wgrad(dkv2, enc16, self.fw('g', lp, 'encoder_attn', 'k_proj', 2), self.acc, dbias=self.fb('g', lp, 'encoder_attn', 'k_proj', 2), dbias_accumulate=self.acc)
            # ---- h1 = LN1(t1), t1 = h_in + out_s(causal_attn(qkv(h_in_b)))
            self._branch_bwd(lp + 'self_attn_layer_norm', k + '.ln1', g('.t1'), dt2, dhb, dt, dtb, lp + 'self_attn.out_proj', 1 + 3 * i)
            ops.linear_dgrad(dtb, self.W(lp + 'self_attn.out_proj.weight'), do)
            self.lin_wgrad(lp + 'self_attn.out_proj', dtb, g('.o1'), has_bias=False)
            q3, dq3 = g('.qkv').view(B, T, 3 * D), dqkv.view(B, T, 3 * D)
            ops.attn_bwd(q3[:, :, 0:D], q3[:, :, D:2 * D], q3[:, :, 2 * D:], g('.o1').view(B, T, D), do.view(B, T, D), g('.lse1'), delta,
                         dq3[:, :, 0:D], dq3[:, :, D:2 * D], dq3[:, :, 2 * D:], H, scale, True, drop=drop, site=200 + 2 * i, q_prescaled=True)
            ops.linear_dgrad(dqkv, self.fw('w', lp, 'self_attn', 'q_proj', 3), dyb)
            ops.linear_wgrad(dqkv, g('.hb'), self.fw('g', lp, 'self_attn', 'q_proj', 3), self.acc, dbias=self.fb('g', lp, 'self_attn', 'q_proj', 3), dbias_accumulate=self.acc)
            dy32 = dt  # residual-path gradient for the layer below (dt now holds d t1)
            if on_layer_done:
                on_layer_done(self.prefix + lp + 'self_attn.q_proj.weight')
        if fused_kv:       # d(encoder output) = [dK_0 | dV_0 | ... | dK_{L-1} | dV_{L-1}] . [W_0; ...; W_{L-1}]: ONE GEMM, fp32 output written once
            ops.linear_dgrad(dkv_all, Tb[tg + 'kv.w_all'], denc, EPI_F32)
        # ---- h0 = LN(emb)
        demb = self.buf('demb', (M, D), F32)
        if self._hidden_drop():        # the mask of site 0 on both gradient streams arriving at the embedding LayerNorm output
            ops.dropout(dy32, dy32, self.drop, 0)
            ops.dropout(dyb, dyb, self.drop, 0)
        self.ln_bwd(dp + 'layernorm_embedding', 'ln_emb', Tb[tg + 'emb'], dy32, dyb, demb, False, None)
        ops.embed_bwd(self.ids, demb, self.G(dp + 'embed_tokens.weight'), self.G(dp + 'embed_positions.weight'), 2, True)
        if on_layer_done:
            on_layer_done(self.prefix + dp + 'embed_tokens.weight')
