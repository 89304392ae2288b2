"""CPU oracle for the Cruller pretrain step -- TEST INFRASTRUCTURE ONLY.

This file is a plain-torch (CPU, no timm / no transformers import) restatement of the
arithmetic executed by the reference's hot path.  Only ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg may import it; the product package
(``pixparse_amd``) never does and fails loudly when its HIP library is missing.

What it restates (reference file:line, relative to /root/reference unless absolute):

* ``Cruller.forward``                      src/pixparse/models/cruller.py:14-21
* image encoder (timm ViT / CLIP-ViT / Swin built by
  ``timm.create_model(name, num_classes=0, global_pool='', img_size=...)``)
                                           src/pixparse/models/image_encoder_timm.py:7-42
  timm is NOT vendored in the reference and NOT installed here; the published timm
  algorithm is restated and cross-checked against HF ViTModel / CLIPVisionModel / SwinModel
  (transformers 5.15.0), see tests/golden/make_golden.py.   Encoder parity vs timm itself:
  **parity unpinned** (no timm source, no reference tests).
* text decoder = ``transformers.BartForCausalLM`` with ``add_cross_attention``
                                           src/pixparse/models/text_decoder_hf.py:10-37,80-103
  /usr/local/lib/python3.10/dist-packages/transformers/models/bart/modeling_bart.py
  :74-111 (learned positions, offset 2), :143-257 (attention), :311-390 (decoder layer, post-LN),
  :552-676 (decoder), :1223-1312 (tied LM head).  Pinned against the live class (golden G1/G2).
* loss / token shift / accumulation       src/pixparse/task/task_cruller_pretrain.py:236-257
* AdamW(eps=1e-6, wd=0) / cosine+warmup / clip-norm / GradScaler semantics
                                           src/pixparse/task/task_cruller_pretrain.py:191-224,259-295
* ``preprocess_ocr_anno`` target masking   src/pixparse/data/preprocess.py:43-110

Precision policies
  ``fp32``  : everything in float32.
  ``bf16``  : the *CUDA* autocast(bfloat16) policy written out with explicit casts
              (SURVEY App. A.6): Linear / conv / attention consume and produce bf16 with fp32
              accumulation, LayerNorm / softmax / cross-entropy compute in fp32, the residual
              stream stays fp32, GELU runs on the bf16 tensor (fp32 internally, bf16 result).
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
BF16 = torch.bfloat16

# --------------------------------------------------------------------------------------
# architecture tables (published timm / HF hyper-parameters)
# --------------------------------------------------------------------------------------
VIT_ARCHS = {
    # timm name -> hyper-parameters
    'vit_base_patch16_224': dict(patch=16, dim=768, depth=12, heads=12, mlp_ratio=4, ln_eps=1e-6,
                                 pre_norm=False, mean=(0.5, 0.5, 0.5), std=(0.5, 0.5, 0.5)),
    'vit_large_patch14_clip_224.datacompxl': dict(
        patch=14, dim=1024, depth=24, heads=16, mlp_ratio=4, ln_eps=1e-5, pre_norm=True,
        mean=(0.48145466, 0.4578275, 0.40821073), std=(0.26862954, 0.26130258, 0.27577711)),
}
SWIN_ARCHS = {
    'swin_tiny_patch4_window7_224': dict(patch=4, embed_dim=96, depths=(2, 2, 6, 2), heads=(3, 6, 12, 24),
                                         window=7, mlp_ratio=4, ln_eps=1e-5,
                                         mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225)),
}
BART_ARCHS = {
    'facebook/bart-base': dict(d_model=768, heads=12, ffn=3072, ln_eps=1e-5),
    'facebook/bart-large': dict(d_model=1024, heads=16, ffn=4096, ln_eps=1e-5),
}


# --------------------------------------------------------------------------------------
# primitive ops under the two precision policies
# --------------------------------------------------------------------------------------
def _linear(x: Tensor, w: Tensor, b: Optional[Tensor], policy: str) -> Tensor:
    """autocast(bf16): inputs/weights cast to bf16, fp32 accumulate, bf16 result."""
    if policy == 'fp32':
        return F.linear(x, w, b)
    return F.linear(x.to(BF16), w.to(BF16), None if b is None else b.to(BF16))


def _layer_norm(x: Tensor, w: Tensor, b: Tensor, eps: float) -> Tensor:
    """autocast keeps layer_norm in fp32 (fp32 in -> fp32 out)."""
    return F.layer_norm(x.float(), (x.shape[-1],), w.float(), b.float(), eps)


def _gelu(x: Tensor) -> Tensor:
    """exact (erf) GELU in the input dtype; bf16 kernels compute in fp32 and round once."""
    if x.dtype == BF16:
        return F.gelu(x.float()).to(BF16)
    return F.gelu(x)


def _attention(q: Tensor, k: Tensor, v: Tensor, scale: float, causal: bool, policy: str,
               bias: Optional[Tensor] = None, fast: bool = False, pdrop=None) -> Tensor:
    """softmax(q k^T * scale + bias) v on [B, H, N, d] tensors.

    bf16 policy mirrors a flash kernel: scores/softmax in fp32 from bf16 operands, the
    probabilities are rounded to bf16 before P.V, the output is rounded to bf16.
    ``fast`` switches to torch's fused CPU SDPA (used only for timing large shapes).
    ``pdrop(probabilities [B, H, Nq, Nk])``: attention-probability dropout (hf BartAttention: softmax, dropout, then P.V); the caller
    supplies the mask (tests: the GPU kernels' own keep mask, crl_attn_dropout_mask).
    """
    if fast and bias is None and pdrop is None:
        return F.scaled_dot_product_attention(q, k, v, is_causal=causal, scale=scale)
    qf, kf, vf = q.float(), k.float(), v.float()
    s = torch.matmul(qf, kf.transpose(-1, -2)) * scale
    if bias is not None:
        s = s + bias.float()
    if causal:
        nq, nk = s.shape[-2], s.shape[-1]
        mask = torch.ones(nq, nk, dtype=torch.bool).tril(diagonal=nk - nq)
        s = s.masked_fill(~mask, float('-inf'))
    p = torch.softmax(s, dim=-1)
    if pdrop is not None:
        p = pdrop(p)
    if policy == 'bf16':
        p = p.to(BF16).float()
    o = torch.matmul(p, vf)
    return o.to(q.dtype)


# --------------------------------------------------------------------------------------
# ViT (timm VisionTransformer)
# --------------------------------------------------------------------------------------
def vit_grid(arch: dict, img_size: Tuple[int, int]) -> Tuple[int, int]:
    return img_size[0] // arch['patch'], img_size[1] // arch['patch']


def vit_forward(p: Dict[str, Tensor], arch: dict, image: Tensor, policy: str = 'bf16',
                prefix: str = '', fast_attn: bool = False) -> Tensor:
    """timm VisionTransformer.forward_features with num_classes=0, global_pool=''.

    image [B, C, H, W] fp32 -> all tokens (cls included) [B, gh*gw+1, D] fp32.
    The conv patch-embed has no padding, so only the top-left (gh*P, gw*P) pixels are read.
    """
    P, D, H = arch['patch'], arch['dim'], arch['heads']
    g = lambda n: p[prefix + n]
    B = image.shape[0]
    w = g('patch_embed.proj.weight')
    b = p.get(prefix + 'patch_embed.proj.bias')
    if policy == 'bf16':
        x = F.conv2d(image.to(BF16), w.to(BF16), None if b is None else b.to(BF16), stride=P)
    else:
        x = F.conv2d(image, w, b, stride=P)
    x = x.flatten(2).transpose(1, 2)  # [B, gh*gw, D], grid row-major (h outer, w inner)
    cls = g('cls_token').expand(B, -1, -1)
    x = torch.cat([cls.float(), x.float()], dim=1)  # cat promotes to fp32
    x = x + g('pos_embed').float()
    if arch['pre_norm']:
        x = _layer_norm(x, g('norm_pre.weight'), g('norm_pre.bias'), arch['ln_eps'])
    N = x.shape[1]
    d = D // H
    for i in range(arch['depth']):
        bp = f'blocks.{i}.'
        h = _layer_norm(x, g(bp + 'norm1.weight'), g(bp + 'norm1.bias'), arch['ln_eps'])
        qkv = _linear(h, g(bp + 'attn.qkv.weight'), g(bp + 'attn.qkv.bias'), policy)
        qkv = qkv.reshape(B, N, 3, H, d).permute(2, 0, 3, 1, 4)
        o = _attention(qkv[0], qkv[1], qkv[2], d ** -0.5, False, policy, fast=fast_attn)
        o = o.transpose(1, 2).reshape(B, N, D)
        x = x + _linear(o, g(bp + 'attn.proj.weight'), g(bp + 'attn.proj.bias'), policy).float()
        h = _layer_norm(x, g(bp + 'norm2.weight'), g(bp + 'norm2.bias'), arch['ln_eps'])
        h = _gelu(_linear(h, g(bp + 'mlp.fc1.weight'), g(bp + 'mlp.fc1.bias'), policy))
        x = x + _linear(h, g(bp + 'mlp.fc2.weight'), g(bp + 'mlp.fc2.bias'), policy).float()
    return _layer_norm(x, g('norm.weight'), g('norm.bias'), arch['ln_eps'])


def vit_param_shapes(arch: dict, in_chans: int, img_size: Tuple[int, int]) -> Dict[str, Tuple[int, ...]]:
    P, D = arch['patch'], arch['dim']
    gh, gw = vit_grid(arch, img_size)
    F_ = D * arch['mlp_ratio']
    s = {'cls_token': (1, 1, D), 'pos_embed': (1, gh * gw + 1, D),
         'patch_embed.proj.weight': (D, in_chans, P, P)}
    if not arch['pre_norm']:
        s['patch_embed.proj.bias'] = (D,)
    else:
        s['norm_pre.weight'] = (D,)
        s['norm_pre.bias'] = (D,)
    for i in range(arch['depth']):
        bp = f'blocks.{i}.'
        s.update({bp + 'norm1.weight': (D,), bp + 'norm1.bias': (D,),
                  bp + 'attn.qkv.weight': (3 * D, D), bp + 'attn.qkv.bias': (3 * D,),
                  bp + 'attn.proj.weight': (D, D), bp + 'attn.proj.bias': (D,),
                  bp + 'norm2.weight': (D,), bp + 'norm2.bias': (D,),
                  bp + 'mlp.fc1.weight': (F_, D), bp + 'mlp.fc1.bias': (F_,),
                  bp + 'mlp.fc2.weight': (D, F_), bp + 'mlp.fc2.bias': (D,)})
    s['norm.weight'] = (D,)
    s['norm.bias'] = (D,)
    return s


# --------------------------------------------------------------------------------------
# Swin (timm SwinTransformer)
# --------------------------------------------------------------------------------------
def swin_relative_position_index(w: int) -> Tensor:
    """[w*w, w*w] index into the [(2w-1)^2, heads] bias table (HF modeling_swin.py:343-370)."""
    coords = torch.stack(torch.meshgrid(torch.arange(w), torch.arange(w), indexing='ij'))
    cf = coords.flatten(1)
    rel = (cf[:, :, None] - cf[:, None, :]).permute(1, 2, 0).contiguous()
    rel[:, :, 0] += w - 1
    rel[:, :, 1] += w - 1
    rel[:, :, 0] *= 2 * w - 1
    return rel.sum(-1)


def swin_shift_mask(Hf: int, Wf: int, w: int, s: int) -> Optional[Tensor]:
    """[nW, w*w, w*w] additive mask, 0 / -100 between different shift regions
    (HF modeling_swin.py:584-607)."""
    if s == 0:
        return None
    img = torch.zeros(1, Hf, Wf, 1)
    cnt = 0
    for hs in (slice(0, -w), slice(-w, -s), slice(-s, None)):
        for ws in (slice(0, -w), slice(-w, -s), slice(-s, None)):
            img[:, hs, ws, :] = cnt
            cnt += 1
    mw = img.view(1, Hf // w, w, Wf // w, w, 1).permute(0, 1, 3, 2, 4, 5).reshape(-1, w * w)
    m = mw[:, None, :] - mw[:, :, None]
    return m.masked_fill(m != 0, -100.0).masked_fill(m == 0, 0.0)


def swin_stage_geometry(arch: dict, img_size: Tuple[int, int]):
    """per stage: (Hf, Wf, C, heads, window, depth); window clamps to the feature size."""
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


def swin_forward(p: Dict[str, Tensor], arch: dict, image: Tensor, policy: str = 'bf16',
                 prefix: str = '', path=None) -> Tensor:
    """timm SwinTransformer.forward_features, flattened to [B, H*W/32^2, 8C].  Drop-path (timm DropPath on both residual branches,
    train mode) is off unless `path(site, branch_output)` is given: site = 2 * global block index + {0: attention, 1: MLP}; the
    caller supplies the per-sample scales (tests: the GPU's own, crl_droppath_scale)."""
    if path is None:
        path = lambda site, t: t
    gj = 0
    g = lambda n: p[prefix + n]
    eps = arch['ln_eps']
    B = image.shape[0]
    P = arch['patch']
    w0, b0 = g('patch_embed.proj.weight'), g('patch_embed.proj.bias')
    if policy == 'bf16':
        x = F.conv2d(image.to(BF16), w0.to(BF16), b0.to(BF16), stride=P)
    else:
        x = F.conv2d(image, w0, b0, stride=P)
    x = x.permute(0, 2, 3, 1)  # NHWC
    x = _layer_norm(x, g('patch_embed.norm.weight'), g('patch_embed.norm.bias'), eps)
    geo = swin_stage_geometry(arch, image.shape[-2:])
    for si, (Hf, Wf, C, heads, w, depth) in enumerate(geo):
        sp = f'layers.{si}.'
        if si > 0:
            # PatchMerging: concat order [(0,0),(1,0),(0,1),(1,1)] -> LN(4C) -> Linear(4C->2C, no bias)
            Cp = C // 2
            x = x.reshape(B, Hf, 2, Wf, 2, Cp).permute(0, 1, 3, 4, 2, 5).flatten(3)
            x = _layer_norm(x, g(sp + 'downsample.norm.weight'), g(sp + 'downsample.norm.bias'), eps)
            x = _linear(x, g(sp + 'downsample.reduction.weight'), None, policy).float()
        d = C // heads
        nW = (Hf // w) * (Wf // w)
        rel_index = swin_relative_position_index(w)
        for bi in range(depth):
            bp = sp + f'blocks.{bi}.'
            shift = 0 if (bi % 2 == 0 or w >= min(Hf, Wf)) else w // 2
            shortcut = x
            h = _layer_norm(x, g(bp + 'norm1.weight'), g(bp + 'norm1.bias'), eps)
            if shift:
                h = torch.roll(h, shifts=(-shift, -shift), dims=(1, 2))
            hw = h.view(B, Hf // w, w, Wf // w, w, C).permute(0, 1, 3, 2, 4, 5).reshape(B * nW, w * w, C)
            qkv = _linear(hw, g(bp + 'attn.qkv.weight'), g(bp + 'attn.qkv.bias'), policy)
            qkv = qkv.reshape(B * nW, w * w, 3, heads, d).permute(2, 0, 3, 1, 4)
            table = g(bp + 'attn.relative_position_bias_table')
            bias = table[rel_index.view(-1)].view(w * w, w * w, heads).permute(2, 0, 1).unsqueeze(0).float()
            mask = swin_shift_mask(Hf, Wf, w, shift)
            if mask is not None:
                bias = bias + mask.repeat(B, 1, 1).unsqueeze(1)
            o = _attention(qkv[0], qkv[1], qkv[2], d ** -0.5, False, policy, bias=bias)
            o = o.transpose(1, 2).reshape(B * nW, w * w, C)
            o = _linear(o, g(bp + 'attn.proj.weight'), g(bp + 'attn.proj.bias'), policy)
            o = o.view(B, Hf // w, Wf // w, w, w, C).permute(0, 1, 3, 2, 4, 5).reshape(B, Hf, Wf, C)
            if shift:
                o = torch.roll(o, shifts=(shift, shift), dims=(1, 2))
            x = shortcut + path(2 * gj, o).float()
            h = _layer_norm(x, g(bp + 'norm2.weight'), g(bp + 'norm2.bias'), eps)
            h = _gelu(_linear(h, g(bp + 'mlp.fc1.weight'), g(bp + 'mlp.fc1.bias'), policy))
            x = x + path(2 * gj + 1, _linear(h, g(bp + 'mlp.fc2.weight'), g(bp + 'mlp.fc2.bias'), policy)).float()
            gj += 1
    x = _layer_norm(x, g('norm.weight'), g('norm.bias'), eps)
    return x.flatten(1, 2)  # NHWC -> [B, HW, C] (the reference's "# flatten?" TODO)


def swin_param_shapes(arch: dict, in_chans: int, img_size: Tuple[int, int]) -> Dict[str, Tuple[int, ...]]:
    P, C0 = arch['patch'], arch['embed_dim']
    s = {'patch_embed.proj.weight': (C0, in_chans, P, P), 'patch_embed.proj.bias': (C0,),
         'patch_embed.norm.weight': (C0,), 'patch_embed.norm.bias': (C0,)}
    for si, (Hf, Wf, C, heads, w, depth) in enumerate(swin_stage_geometry(arch, img_size)):
        sp = f'layers.{si}.'
        if si > 0:
            s[sp + 'downsample.norm.weight'] = (2 * C,)
            s[sp + 'downsample.norm.bias'] = (2 * C,)
            s[sp + 'downsample.reduction.weight'] = (C, 2 * C)
        F_ = C * arch['mlp_ratio']
        for bi in range(depth):
            bp = sp + f'blocks.{bi}.'
            s.update({bp + 'norm1.weight': (C,), bp + 'norm1.bias': (C,),
                      bp + 'attn.qkv.weight': (3 * C, C), bp + 'attn.qkv.bias': (3 * C,),
                      bp + 'attn.relative_position_bias_table': ((2 * w - 1) ** 2, heads),
                      bp + 'attn.proj.weight': (C, C), bp + 'attn.proj.bias': (C,),
                      bp + 'norm2.weight': (C,), bp + 'norm2.bias': (C,),
                      bp + 'mlp.fc1.weight': (F_, C), bp + 'mlp.fc1.bias': (F_,),
                      bp + 'mlp.fc2.weight': (C, F_), bp + 'mlp.fc2.bias': (C,)})
    Cl = C0 * 2 ** (len(arch['depths']) - 1)
    s['norm.weight'] = (Cl,)
    s['norm.bias'] = (Cl,)
    return s


# --------------------------------------------------------------------------------------
# BART decoder with cross attention + tied LM head
# --------------------------------------------------------------------------------------
def bart_decoder_forward(p: Dict[str, Tensor], arch: dict, n_layers: int, input_ids: Tensor,
                         enc: Tensor, policy: str = 'bf16', prefix: str = '',
                         fast_attn: bool = False, drop=None) -> Tensor:
    """BartForCausalLM(input_ids, encoder_hidden_states=enc).logits  [B, T, V].

    Pure causal self-attention (the reference passes no attention_mask), unmasked cross
    attention, post-LN layers, learned positions with offset 2, embed scale 1.0.
    Dropout (modeling_bart.py:362,377,384-386,654; live in the reference only for a train-mode decoder) is off unless
    `drop(site, tensor)` is given: it is applied to the embedding LayerNorm output (site 0) and to the three branch outputs of
    layer i before their residual joins (sites 1 + 3 i + {0: self-attention, 1: cross attention, 2: fc2}) -- the hidden-state
    dropout sites of BartDecoder / BartDecoderLayer; the caller supplies the mask (tests: the GPU kernel's own Philox mask).
    bart-base additionally drops the FFN activation behind the GELU (`drop.act(300 + i, activation)`, hf:384) and the attention
    probabilities (`drop.attn(site, probabilities)`, sites 200 + 2 i: self, 201 + 2 i: cross) when the callable carries those attributes.
    """
    if drop is None:
        drop = lambda site, t: t
    attn_drop = getattr(drop, 'attn', None)
    act_drop = getattr(drop, 'act', None) or (lambda site, t: t)
    pd = (lambda site: (lambda pr: attn_drop(site, pr))) if attn_drop is not None else (lambda site: None)
    D, H = arch['d_model'], arch['heads']
    d = D // H
    eps = arch['ln_eps']
    g = lambda n: p[prefix + n]
    dp = 'model.decoder.'
    B, T = input_ids.shape
    S = enc.shape[1]
    h = g(dp + 'embed_tokens.weight')[input_ids].float()
    h = h + g(dp + 'embed_positions.weight')[torch.arange(T) + 2].float()
    h = drop(0, _layer_norm(h, g(dp + 'layernorm_embedding.weight'), g(dp + 'layernorm_embedding.bias'), eps))

    def heads(t, n):
        return t.view(B, n, H, d).transpose(1, 2)

    for i in range(n_layers):
        lp = dp + f'layers.{i}.'
        # self attention (causal)
        q = _linear(h, g(lp + 'self_attn.q_proj.weight'), g(lp + 'self_attn.q_proj.bias'), policy)
        k = _linear(h, g(lp + 'self_attn.k_proj.weight'), g(lp + 'self_attn.k_proj.bias'), policy)
        v = _linear(h, g(lp + 'self_attn.v_proj.weight'), g(lp + 'self_attn.v_proj.bias'), policy)
        o = _attention(heads(q, T), heads(k, T), heads(v, T), d ** -0.5, True, policy, fast=fast_attn, pdrop=pd(200 + 2 * i))
        o = o.transpose(1, 2).reshape(B, T, D)
        o = drop(1 + 3 * i, _linear(o, g(lp + 'self_attn.out_proj.weight'), g(lp + 'self_attn.out_proj.bias'), policy))
        h = _layer_norm(h + o.float(), g(lp + 'self_attn_layer_norm.weight'),
                        g(lp + 'self_attn_layer_norm.bias'), eps)
        # cross attention (K/V of the encoder states recomputed in every layer)
        q = _linear(h, g(lp + 'encoder_attn.q_proj.weight'), g(lp + 'encoder_attn.q_proj.bias'), policy)
        k = _linear(enc, g(lp + 'encoder_attn.k_proj.weight'), g(lp + 'encoder_attn.k_proj.bias'), policy)
        v = _linear(enc, g(lp + 'encoder_attn.v_proj.weight'), g(lp + 'encoder_attn.v_proj.bias'), policy)
        o = _attention(heads(q, T), heads(k, S), heads(v, S), d ** -0.5, False, policy, fast=fast_attn, pdrop=pd(201 + 2 * i))
        o = o.transpose(1, 2).reshape(B, T, D)
        o = drop(2 + 3 * i, _linear(o, g(lp + 'encoder_attn.out_proj.weight'), g(lp + 'encoder_attn.out_proj.bias'), policy))
        h = _layer_norm(h + o.float(), g(lp + 'encoder_attn_layer_norm.weight'),
                        g(lp + 'encoder_attn_layer_norm.bias'), eps)
        # FFN
        f = act_drop(300 + i, _gelu(_linear(h, g(lp + 'fc1.weight'), g(lp + 'fc1.bias'), policy)))
        f = drop(3 + 3 * i, _linear(f, g(lp + 'fc2.weight'), g(lp + 'fc2.bias'), policy))
        h = _layer_norm(h + f.float(), g(lp + 'final_layer_norm.weight'), g(lp + 'final_layer_norm.bias'), eps)
    # tied LM head, no bias
    return _linear(h, g(dp + 'embed_tokens.weight'), None, policy)


def bart_param_shapes(arch: dict, n_layers: int, vocab: int, max_pos: int) -> Dict[str, Tuple[int, ...]]:
    D, F_ = arch['d_model'], arch['ffn']
    dp = 'model.decoder.'
    s = {dp + 'embed_tokens.weight': (vocab, D), dp + 'embed_positions.weight': (max_pos + 2, D),
         dp + 'layernorm_embedding.weight': (D,), dp + 'layernorm_embedding.bias': (D,)}
    for i in range(n_layers):
        lp = dp + f'layers.{i}.'
        for a in ('self_attn', 'encoder_attn'):
            for proj in ('q_proj', 'k_proj', 'v_proj', 'out_proj'):
                s[lp + f'{a}.{proj}.weight'] = (D, D)
                s[lp + f'{a}.{proj}.bias'] = (D,)
            s[lp + f'{a}_layer_norm.weight'] = (D,)
            s[lp + f'{a}_layer_norm.bias'] = (D,)
        s[lp + 'fc1.weight'] = (F_, D)
        s[lp + 'fc1.bias'] = (F_,)
        s[lp + 'fc2.weight'] = (D, F_)
        s[lp + 'fc2.bias'] = (D,)
        s[lp + 'final_layer_norm.weight'] = (D,)
        s[lp + 'final_layer_norm.bias'] = (D,)
    return s


# --------------------------------------------------------------------------------------
# Cruller = encoder + decoder, loss, train step
# --------------------------------------------------------------------------------------
class ModelSpec:
    """Resolved architecture of one Cruller config (encoder + decoder + sizes)."""

    def __init__(self, encoder: str, decoder: str, n_layers: int, max_length: int,
                 img_size: Tuple[int, int], in_chans: int, vocab: int = 50267):
        self.encoder, self.decoder = encoder, decoder
        self.n_layers, self.max_length = n_layers, max_length
        self.img_size, self.in_chans, self.vocab = tuple(img_size), in_chans, vocab
        self.enc_kind = 'swin' if encoder in SWIN_ARCHS else 'vit'
        self.enc_arch = SWIN_ARCHS[encoder] if self.enc_kind == 'swin' else VIT_ARCHS[encoder]
        self.dec_arch = BART_ARCHS[decoder]

    def param_shapes(self) -> Dict[str, Tuple[int, ...]]:
        """state_dict names exactly as the reference checkpoint has them (app/train.py:64-67)."""
        fn = swin_param_shapes if self.enc_kind == 'swin' else vit_param_shapes
        s = {'image_encoder.trunk.' + k: v for k, v in fn(self.enc_arch, self.in_chans, self.img_size).items()}
        s.update({'text_decoder.trunk.' + k: v
                  for k, v in bart_param_shapes(self.dec_arch, self.n_layers, self.vocab, self.max_length).items()})
        return s


def init_params(spec: ModelSpec, seed: int = 0, std: float = 0.02) -> Dict[str, Tensor]:
    """Deterministic random init (pretrained weights are unobtainable offline).
    LN weights ~ 1 + N(0, std), everything else N(0, std), all values fp32."""
    gen = torch.Generator().manual_seed(seed)
    out = {}
    for name, shape in spec.param_shapes().items():
        t = torch.randn(*shape, generator=gen) * std
        if ('norm' in name) and name.endswith('.weight'):
            t = t + 1.0
        out[name] = t
    return out


def cruller_forward(p: Dict[str, Tensor], spec: ModelSpec, image: Tensor, text_input: Tensor,
                    policy: str = 'bf16', fast_attn: bool = False, drop=None) -> Tensor:
    """models/cruller.py:14-21 -> logits [B, T, V]."""
    if spec.enc_kind == 'swin':
        enc = swin_forward(p, spec.enc_arch, image, policy, prefix='image_encoder.trunk.', path=getattr(drop, 'path', None))
    else:
        enc = vit_forward(p, spec.enc_arch, image, policy, prefix='image_encoder.trunk.', fast_attn=fast_attn)
    return bart_decoder_forward(p, spec.dec_arch, spec.n_layers, text_input, enc, policy,
                                prefix='text_decoder.trunk.', fast_attn=fast_attn, drop=drop)


def encode_image(p: Dict[str, Tensor], spec: ModelSpec, image: Tensor, policy: str = 'bf16') -> Tensor:
    """model.image_encoder(image) (task_cruller_eval_ocr.py) -> [B, S, D]"""
    if spec.enc_kind == 'swin':
        return swin_forward(p, spec.enc_arch, image, policy, prefix='image_encoder.trunk.')
    return vit_forward(p, spec.enc_arch, image, policy, prefix='image_encoder.trunk.')


def greedy_generate(p: Dict[str, Tensor], spec: ModelSpec, encoder_outputs: Tensor, prompt_id: int, eos_id: int,
                    max_recursion_length: int, policy: str = 'bf16', return_logits: bool = False):
    """utils/ocr_utils.py:165-197 get_generated_tokens with use_sample=False: the WHOLE decoder is re-run on the growing
    sequence for every new token (no KV cache); samples that already produced eos keep being extended until all have;
    the token that completes the last sample is not appended.  prompt_id: one token id, or a multi-token prompt (list of ids /
    [B, P] tensor) as the DocVQA / CORD eval loops start from (task/task_cruller_eval_docvqa.py:279-297: the decoder is re-run on
    `<s_docvqa><s_question>...</s_question><s_answer>` + everything generated so far)."""
    B = encoder_outputs.shape[0]
    if isinstance(prompt_id, int):
        input_ids = torch.full((B, 1), prompt_id, dtype=torch.int64)
    else:
        input_ids = torch.as_tensor(prompt_id, dtype=torch.int64)
        input_ids = input_ids.view(1, -1).expand(B, -1).clone() if input_ids.dim() == 1 else input_ids.clone()
    finished = torch.zeros(B, dtype=torch.bool)
    steps = []
    for _ in range(max_recursion_length):
        logits = bart_decoder_forward(p, spec.dec_arch, spec.n_layers, input_ids, encoder_outputs, policy,
                                      prefix='text_decoder.trunk.')
        nxt = logits[:, -1, :]
        steps.append(nxt.float())
        next_id = torch.argmax(nxt.float(), dim=-1, keepdim=True)
        finished |= next_id.squeeze(-1) == eos_id
        if finished.all():
            break
        input_ids = torch.cat([input_ids, next_id], dim=-1)
    return (input_ids, steps) if return_logits else input_ids


def cross_entropy(logits: Tensor, target: Tensor, ignore_index: int = -100) -> Tensor:
    """nn.CrossEntropyLoss(ignore_index=-100): fp32 log-softmax, mean over non-ignored."""
    return F.cross_entropy(logits.float().view(-1, logits.shape[-1]), target.reshape(-1),
                           ignore_index=ignore_index)


def cruller_loss(p, spec: ModelSpec, image: Tensor, text_input: Tensor, text_target: Tensor,
                 policy: str = 'bf16', accum_steps: int = 1, fast_attn: bool = False, drop=None) -> Tensor:
    """the `_forward` closure, task_cruller_pretrain.py:247-257 (inputs already shifted)."""
    logits = cruller_forward(p, spec, image, text_input, policy, fast_attn, drop)
    loss = cross_entropy(logits, text_target)
    if accum_steps > 1:
        loss = loss / accum_steps
    return loss


def classifier_loss(p: Dict[str, Tensor], spec: 'ModelSpec', image: Tensor, label: Tensor, policy: str = 'bf16') -> Tuple[Tensor, Tensor]:
    """the classification fine-tune's forward (ref task/task_cruller_finetune_xent.py:146-151, :237-247): nn.Sequential(image encoder,
    GetCLSToken = x[:, 0, :], nn.Linear(features, classes)) under autocast, CrossEntropyLoss(ignore_index=-100) -> (loss, logits)"""
    assert spec.enc_kind == 'vit', 'token 0 is a class token only for the ViT encoders'
    feats = vit_forward(p, spec.enc_arch, image, policy, prefix='image_encoder.trunk.')[:, 0, :]
    logits = _linear(feats, p['final_fc.weight'], p['final_fc.bias'], policy)
    return cross_entropy(logits, label), logits


def shift_tokens(text_input: Tensor, text_target: Tensor) -> Tuple[Tensor, Tensor]:
    """task_cruller_pretrain.py:241-242."""
    return text_input[:, :-1], text_target[:, 1:]


# --------------------------------------------------------------------------------------
# optimiser / schedule / clip
# --------------------------------------------------------------------------------------
def cosine_lr(t: int, base_lr: float, warmup_t: int, t_initial: int, warmup_lr_init: float = 0.0,
              lr_min: float = 0.0) -> float:
    """timm CosineLRScheduler.step_update(t) with t_in_epochs=False, warmup_prefix=False, cycle_limit=1."""
    if t < warmup_t:
        return warmup_lr_init + t * (base_lr - warmup_lr_init) / warmup_t
    if t < t_initial:
        return lr_min + 0.5 * (base_lr - lr_min) * (1 + math.cos(math.pi * t / t_initial))
    return lr_min


def clip_grad_norm_(grads, max_norm: float) -> Tensor:
    """torch.nn.utils.clip_grad_norm_(norm_type=2): scale = max_norm/(total+1e-6) clamped to 1."""
    total = torch.sqrt(sum((g.float() ** 2).sum() for g in grads))
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    for g in grads:
        g.mul_(coef)
    return total


def adamw_step(p: Tensor, g: Tensor, m: Tensor, v: Tensor, step: int, lr: float,
               beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-6, weight_decay: float = 0.0):
    """torch.optim.AdamW single-tensor update, in place; ``step`` counts from 1."""
    if weight_decay:
        p.mul_(1 - lr * weight_decay)
    m.mul_(beta1).add_(g, alpha=1 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-lr / bc1)


class OracleTrainer:
    """One rank of the reference train loop over the restated model: forward, loss, backward,
    (clip) AdamW, cosine LR.  GradScaler with bf16 is a power-of-two scale that cancels exactly,
    so it is modelled as identity."""

    def __init__(self, spec: ModelSpec, params: Dict[str, Tensor], lr=5e-4, betas=(0.9, 0.999), eps=1e-6,
                 clip_grad: Optional[float] = None, accum_steps: int = 1, warmup_t: int = 0,
                 t_initial: int = 1000, policy: str = 'bf16', fast_attn: bool = False):
        self.spec, self.policy, self.fast_attn = spec, policy, fast_attn
        self.params = {k: v.clone().requires_grad_(True) for k, v in params.items()}
        self.m = {k: torch.zeros_like(v) for k, v in params.items()}
        self.v = {k: torch.zeros_like(v) for k, v in params.items()}
        self.base_lr, self.betas, self.eps = lr, betas, eps
        self.clip_grad, self.accum_steps = clip_grad, accum_steps
        self.warmup_t, self.t_initial = warmup_t, t_initial
        self.step = 0
        self.micro = 0
        self.lr = cosine_lr(0, lr, warmup_t, t_initial)
        self.last_grad_norm = None

    def train_step(self, sample) -> float:
        image, text, target = sample
        ti, tt = shift_tokens(text, target)
        loss = cruller_loss(self.params, self.spec, image, ti, tt, self.policy, self.accum_steps, self.fast_attn)
        loss.backward()
        self.micro += 1
        if self.micro % self.accum_steps:
            return float(loss)
        with torch.no_grad():
            grads = [p.grad for p in self.params.values()]
            if self.clip_grad is not None:
                self.last_grad_norm = float(clip_grad_norm_(grads, self.clip_grad))
            self.step += 1
            for k, p in self.params.items():
                adamw_step(p, p.grad, self.m[k], self.v[k], self.step, self.lr, self.betas[0], self.betas[1], self.eps)
                p.grad = None
            self.lr = cosine_lr(self.step, self.base_lr, self.warmup_t, self.t_initial)
        return float(loss)


# --------------------------------------------------------------------------------------
# data contract (src/pixparse/data/preprocess.py:43-110)
# --------------------------------------------------------------------------------------
def make_targets(text: Tensor, pad_token_id: int, prompt_end_token_id: int, ignore_id: int = -100) -> Tensor:
    """target = text.clone(); pad -> ignore; prefix up to and including the prompt-end token ->
    ignore.  Keeps the reference quirk that ``torch.nonzero(...).sum()`` SUMS the matching indices."""
    target = text.clone()
    target[target == pad_token_id] = ignore_id
    target[:torch.nonzero(target == prompt_end_token_id).sum() + 1] = ignore_id
    return target


def synthetic_sample(spec: ModelSpec, batch: int, seed: int = 42, rank: int = 0, ragged: bool = False):
    """SURVEY §8d synthetic inputs: image ~ N(0,1); tokens uniform in [3, 50265) with
    <s_pretrain> (= vocab-1) first and eos (2) last; target = text with the prompt masked.
    ``ragged`` puts eos at a random length in [L/2, L) and pads (id 1 / -100) after it."""
    gen = torch.Generator().manual_seed(seed + rank)
    H, W = spec.img_size
    L = spec.max_length
    image = torch.randn(batch, spec.in_chans, H, W, generator=gen)
    hi = min(50265, spec.vocab - 2)
    tokens = torch.randint(3, hi, (batch, L), generator=gen)
    tokens[:, 0] = spec.vocab - 1
    tokens[:, L - 1] = 2
    if ragged:
        for b in range(batch):
            n = int(torch.randint(L // 2, L, (1,), generator=gen))
            tokens[b, n] = 2
            tokens[b, n + 1:] = 1
    target = torch.stack([make_targets(t, 1, spec.vocab - 1) for t in tokens])
    return image, tokens, target
