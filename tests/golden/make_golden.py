#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ (run in the build container only).

Sources of truth used here (none of them travels to the GPU box; only the vectors do):
  * transformers 5.15.0 ``BartForCausalLM`` built exactly as
    /root/reference/src/pixparse/models/text_decoder_hf.py:10-37 builds it (add_cross_attention,
    decoder_layers, max_position_embeddings, then resize_token_embeddings) -> G1, G2
  * /root/reference/src/pixparse/data/preprocess.py executed standalone with a stub tokenizer -> G3
  * HF ``ViTModel`` / ``CLIPVisionModel`` / ``SwinModel`` as stand-ins for the (absent) timm
    encoders, weights renamed to timm's state-dict layout -> G4
  * ``torch.optim.AdamW`` / ``torch.nn.utils.clip_grad_norm_`` and the closed-form cosine
    schedule -> G5

Usage:  python tests/golden/make_golden.py   (writes *.safetensors / *.json next to this file)
"""
import importlib.util
import json
import math
import os
import random
import sys

import torch
from safetensors.torch import save_file

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = '/root/reference/src/pixparse'

torch.manual_seed(0)


def bf16_exact(t):
    return t.to(torch.bfloat16).float()


# ------------------------------------------------------------------ G1 / G2: decoder
def build_bart(d_model, heads, ffn, layers, vocab, max_len):
    import transformers
    cfg = transformers.BartConfig(
        vocab_size=vocab - 2, d_model=d_model, decoder_attention_heads=heads, decoder_ffn_dim=ffn,
        encoder_layers=1, encoder_attention_heads=heads, encoder_ffn_dim=ffn,
        dropout=0.0, attention_dropout=0.0, activation_dropout=0.0)
    # --- the three lines of text_decoder_hf.py:14-22
    cfg.add_cross_attention = True
    cfg.decoder_layers = layers
    cfg.max_position_embeddings = max_len
    model = transformers.AutoModelForCausalLM.from_config(cfg)
    model.resize_token_embeddings(vocab)  # task_cruller_pretrain.py:115-116 (+2 special tokens)
    model.eval()
    # randomise every parameter (default init leaves biases at 0 and LN at 1)
    gen = torch.Generator().manual_seed(1234)
    with torch.no_grad():
        for n, p in model.named_parameters():
            t = torch.randn(p.shape, generator=gen) * (0.08 if p.dim() > 1 else 0.05)
            if 'layer_norm' in n or 'layernorm' in n:
                if n.endswith('weight'):
                    t = t + 1.0
            p.copy_(bf16_exact(t))
    assert model.lm_head.weight.data_ptr() == model.model.decoder.embed_tokens.weight.data_ptr()
    return model


def gen_decoder(tag, d_model, heads, ffn, layers, vocab, T, S, B, logit_cols=None, all_grads=True):
    model = build_bart(d_model, heads, ffn, layers, vocab, T + 1)
    gen = torch.Generator().manual_seed(7)
    ids = torch.randint(0, vocab, (B, T), generator=gen)
    enc = bf16_exact(torch.randn(B, S, d_model, generator=gen))
    target = torch.randint(0, vocab, (B, T), generator=gen)
    target[:, 0] = -100
    target[1, T // 2:] = -100  # ragged tail, like padding
    enc_req = enc.clone().requires_grad_(True)
    out = model(input_ids=ids, encoder_hidden_states=enc_req, return_dict=True, use_cache=False)
    logits = out['logits']
    loss = torch.nn.CrossEntropyLoss(ignore_index=-100)(logits.view(-1, vocab), target.view(-1))
    loss.backward()
    with torch.no_grad(), torch.autocast('cpu', dtype=torch.bfloat16):
        logits_bf16 = model(input_ids=ids, encoder_hidden_states=enc, return_dict=True, use_cache=False)['logits'].float()
        loss_bf16 = torch.nn.functional.cross_entropy(logits_bf16.view(-1, vocab), target.view(-1), ignore_index=-100)
    tensors = {'in.input_ids': ids, 'in.enc': enc, 'in.target': target}
    sd = model.state_dict()
    for k, v in sd.items():
        if k == 'lm_head.weight':
            continue  # tied
        tensors['w.' + k] = v.detach().to(torch.bfloat16).contiguous()
    lc = logit_cols or vocab
    tensors['out.logits_fp32'] = logits.detach()[:, :, :lc].contiguous()
    tensors['out.logits_bf16'] = logits_bf16[:, :, :lc].contiguous()
    tensors['out.loss_fp32'] = loss.detach().reshape(1)
    tensors['out.loss_bf16'] = loss_bf16.reshape(1)
    tensors['out.grad_enc'] = enc_req.grad.detach()
    gnorm = {}
    for n, p in model.named_parameters():
        if n == 'lm_head.weight':
            continue
        gnorm[n] = float(p.grad.norm())
        if all_grads:
            tensors['g.' + n] = p.grad.detach().contiguous()
    meta = dict(d_model=d_model, heads=heads, ffn=ffn, layers=layers, vocab=vocab, T=T, S=S, B=B,
                logit_cols=lc, grad_norms=gnorm)
    save_file(tensors, os.path.join(HERE, f'{tag}.safetensors'))
    with open(os.path.join(HERE, f'{tag}.json'), 'w') as f:
        json.dump(meta, f, indent=1)
    print(tag, 'loss fp32', float(loss), 'bf16', float(loss_bf16))


# ------------------------------------------------------------------ G3: preprocess
class StubTokenizer:
    """char-level tokenizer with BART's special ids: pad=1, eos=2, <s_pretrain>=50266."""
    pad_token_id = 1
    eos_token = '</s>'
    specials = {'</s>': 2, '<s_pretrain>': 50266, '<sep/>': 50265}

    def convert_tokens_to_ids(self, tok):
        return self.specials[tok]

    def __call__(self, text, add_special_tokens=False, return_tensors='pt', max_length=None,
                 padding='max_length', truncation=True):
        ids = []
        i = 0
        while i < len(text):
            for s, sid in self.specials.items():
                if text.startswith(s, i):
                    ids.append(sid)
                    i += len(s)
                    break
            else:
                ids.append(ord(text[i]) + 100)
                i += 1
        ids = ids[:max_length]
        ids = ids + [self.pad_token_id] * (max_length - len(ids))

        class R:
            pass
        r = R()
        r.input_ids = torch.tensor([ids])
        return r


def gen_preprocess():
    spec = importlib.util.spec_from_file_location('ref_preprocess', os.path.join(REF, 'data/preprocess.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    tok = StubTokenizer()
    cases = []
    annos = [
        {'pages': [{'text': []}, {'text': ['ab', 'c']}]},
        {'pages': [{'text': ['hello world', 'second line', 'third']}]},
        {'pages': [{'text': ['x' * 40]}]},  # truncation: eos cut off
    ]
    for L in (12, 24):
        for ai, anno in enumerate(annos):
            out, meta = mod.preprocess_ocr_anno(anno, tok, L, '<s_pretrain>', '<s_pretrain>',
                                                generator=random.Random(0))
            cases.append(dict(fn='ocr', L=L, anno=anno, text=out['text'][0].tolist(),
                              target=out['target'][0].tolist(), meta=meta))
        out = mod.preprocess_text_anno('some raw text', tok, L, '<s_pretrain>', '<s_pretrain>')
        cases.append(dict(fn='text', L=L, anno='some raw text', text=out['text'][0].tolist(),
                          target=out['target'][0].tolist()))
    with open(os.path.join(HERE, 'g3_preprocess.json'), 'w') as f:
        json.dump(cases, f, indent=1)
    print('g3', len(cases), 'cases')


# ------------------------------------------------------------------ G6: fine-tune glue (json2token, prompt masking)
def _ref_function(path, name, cls=None, extra_globals=None):
    """compile ONE function (or one method of `cls`) of a reference file without importing the file's other
    dependencies (zss / nltk / timm are not installed here); the function's own body runs unmodified"""
    import ast
    src = open(os.path.join(REF, path)).read()
    tree = ast.parse(src)
    body = tree.body
    if cls is not None:
        body = next(n for n in body if isinstance(n, ast.ClassDef) and n.name == cls).body
    fn = next(n for n in body if isinstance(n, ast.FunctionDef) and n.name == name)
    mod = ast.Module(body=[fn], type_ignores=[])
    g = {'__builtins__': __builtins__, 'torch': torch, 're': __import__('re')}
    g.update(extra_globals or {})
    exec(compile(mod, os.path.join(REF, path), 'exec'), g)
    return g[name]


def gen_finetune():
    json2token = _ref_function('utils/json_utils.py', 'json2token')
    token2json = _ref_function('utils/json_utils.py', 'token2json')
    specials = ['<s>', '</s>', '<pad>', '<sep/>', '<yes/>', '<no/>']
    objs = [
        {'menu': [{'nm': 'latte', 'cnt': '2', 'price': '9.0'}, {'nm': 'tea', 'cnt': '1'}], 'total': {'total_price': '11.0'}},
        {'menu': {'nm': 'single', 'price': 3}, 'sub_total': {'subtotal_price': '3', 'tax_price': ['0.1', '0.2']}},
        {'text_sequence': 'already a sequence'},
        {'answer': 'yes', 'flag': 'no', 'other': 'maybe'},
        ['a', 'b', {'k': 'v'}],
        'leaf',
    ]
    cases = []
    for o in objs:
        for sort_keys in (False, True):
            out = json2token(o, specials, [], True, sort_keys)
            text = out if isinstance(out, str) else out[0]
            toks = [] if isinstance(out, str) else sorted(out[1])
            back = token2json(text, added_vocab={t: i for i, t in enumerate(specials)})
            cases.append(dict(obj=o, sort_json_key=sort_keys, specials=specials, text=text, key_tokens=toks, token2json=back))
    # prompt masking: the method body of the three fine-tune tasks, bound to a stub with their token ids
    from pixparse_amd.tokenizers import ByteBartTokenizer   # byte-level stand-in with BART's special ids
    tok = ByteBartTokenizer()
    tok.add_special_tokens({'additional_special_tokens': ['<sep/>', '<s_pretrain>', '<s_docvqa>', '<s_answer>', '<s_rvlcdip>',
                                                          '<s_question>', '</s_question>', '</s_answer>', '<letter/>']})
    masks = []
    for path, cls, prompt_end, seqs in (
            ('task/task_cruller_finetune_RVLCDIP.py', 'TaskCrullerFinetuneRVLCDIP', '<s_rvlcdip>', [('<s_rvlcdip><letter/></s>', 5)]),
            ('task/task_cruller_finetune_docvqa.py', 'TaskCrullerFinetuneDOCVQA', '<s_answer>',
             [('<s_docvqa><s_question>who?</s_question><s_answer>me</s_answer></s>', 32),
              ('<s_docvqa><s_question>a much longer question text</s_question><s_answer>x</s_answer></s>', 24)]),
            ('task/task_cruller_finetune_CORD.py', 'TaskCrullerFinetuneCORD', '<s_pretrain>', [('<s_pretrain>abc</s>', 8)])):
        fn = _ref_function(path, 'text_input_to_target', cls=cls)

        class _Self:
            pass
        me = _Self()
        me.tokenizer = _Self()
        me.tokenizer.trunk = tok
        me.prompt_end_token = prompt_end
        for text, L in seqs:
            ids = tok(text, add_special_tokens=False, return_tensors='pt', max_length=L, padding='max_length', truncation=True).input_ids[0]
            tgt = fn(me, ids)
            masks.append(dict(task=cls, prompt_end=prompt_end, text=text, max_length=L, ids=ids.tolist(), target=tgt.tolist()))
    with open(os.path.join(HERE, 'g6_finetune.json'), 'w') as f:
        json.dump(dict(json2token=cases, text_input_to_target=masks, added_tokens=tok.added), f, indent=1)
    print('g6', len(cases), 'json2token cases,', len(masks), 'masking cases')



# ------------------------------------------------------------------ G4: encoders
def randomise(model, seed):
    gen = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for n, p in model.named_parameters():
            t = torch.randn(p.shape, generator=gen) * (0.1 if p.dim() > 1 else 0.05)
            if ('norm' in n.lower()) and n.endswith('weight'):
                t = t + 1.0
            p.copy_(t)


def gen_vit():
    import transformers
    H, W, P, D, L, NH = 37, 50, 8, 32, 2, 4  # non-multiple image size: grid floors to 4x6
    cfg = transformers.ViTConfig(hidden_size=D, num_hidden_layers=L, num_attention_heads=NH, intermediate_size=4 * D,
                                 image_size=(H, W), patch_size=P, num_channels=1, layer_norm_eps=1e-6,
                                 hidden_act='gelu', hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    m = transformers.ViTModel(cfg, add_pooling_layer=False).eval()
    randomise(m, 11)
    img = torch.randn(2, 1, H, W, generator=torch.Generator().manual_seed(3))
    with torch.no_grad():
        out = m(pixel_values=img, interpolate_pos_encoding=False).last_hidden_state
    sd = m.state_dict()
    t = {'cls_token': sd['embeddings.cls_token'], 'pos_embed': sd['embeddings.position_embeddings'],
         'patch_embed.proj.weight': sd['embeddings.patch_embeddings.projection.weight'],
         'patch_embed.proj.bias': sd['embeddings.patch_embeddings.projection.bias'],
         'norm.weight': sd['layernorm.weight'], 'norm.bias': sd['layernorm.bias']}
    for i in range(L):
        hp, tp = f'layers.{i}.', f'blocks.{i}.'
        for wb in ('weight', 'bias'):
            t[tp + 'norm1.' + wb] = sd[hp + 'layernorm_before.' + wb]
            t[tp + 'norm2.' + wb] = sd[hp + 'layernorm_after.' + wb]
            t[tp + 'attn.qkv.' + wb] = torch.cat([sd[hp + f'attention.{n}_proj.' + wb] for n in ('q', 'k', 'v')], 0)
            t[tp + 'attn.proj.' + wb] = sd[hp + 'attention.o_proj.' + wb]
            t[tp + 'mlp.fc1.' + wb] = sd[hp + 'mlp.fc1.' + wb]
            t[tp + 'mlp.fc2.' + wb] = sd[hp + 'mlp.fc2.' + wb]
    tensors = {'w.' + k: v.contiguous() for k, v in t.items()}
    tensors['in.image'] = img
    tensors['out.tokens'] = out.contiguous()
    save_file(tensors, os.path.join(HERE, 'g4_vit.safetensors'))
    json.dump(dict(patch=P, dim=D, depth=L, heads=NH, mlp_ratio=4, ln_eps=1e-6, pre_norm=False, img_size=[H, W],
                   in_chans=1), open(os.path.join(HERE, 'g4_vit.json'), 'w'))
    print('g4_vit', tuple(out.shape))


def gen_clip():
    import transformers
    S, P, D, L, NH = 30, 7, 32, 2, 4  # 30 // 7 = 4 -> 4x4 grid, floor-cropped
    cfg = transformers.CLIPVisionConfig(hidden_size=D, intermediate_size=4 * D, num_hidden_layers=L,
                                        num_attention_heads=NH, image_size=S, patch_size=P, num_channels=3,
                                        hidden_act='gelu', layer_norm_eps=1e-5, attention_dropout=0.0)
    m = transformers.CLIPVisionModel(cfg).eval()
    randomise(m, 12)
    img = torch.randn(2, 3, S, S, generator=torch.Generator().manual_seed(4))
    with torch.no_grad():
        hs = m(pixel_values=img).last_hidden_state
        out = m.post_layernorm(hs)  # timm's `norm` applies to ALL tokens
    sd = m.state_dict()
    t = {'cls_token': sd['embeddings.class_embedding'].view(1, 1, D),
         'pos_embed': sd['embeddings.position_embedding.weight'].unsqueeze(0),
         'patch_embed.proj.weight': sd['embeddings.patch_embedding.weight'],
         'norm_pre.weight': sd['pre_layrnorm.weight'], 'norm_pre.bias': sd['pre_layrnorm.bias'],
         'norm.weight': sd['post_layernorm.weight'], 'norm.bias': sd['post_layernorm.bias']}
    for i in range(L):
        hp, tp = f'encoder.layers.{i}.', f'blocks.{i}.'
        for wb in ('weight', 'bias'):
            t[tp + 'norm1.' + wb] = sd[hp + 'layer_norm1.' + wb]
            t[tp + 'norm2.' + wb] = sd[hp + 'layer_norm2.' + wb]
            t[tp + 'attn.qkv.' + wb] = torch.cat([sd[hp + f'self_attn.{n}_proj.' + wb] for n in ('q', 'k', 'v')], 0)
            t[tp + 'attn.proj.' + wb] = sd[hp + 'self_attn.out_proj.' + wb]
            t[tp + 'mlp.fc1.' + wb] = sd[hp + 'mlp.fc1.' + wb]
            t[tp + 'mlp.fc2.' + wb] = sd[hp + 'mlp.fc2.' + wb]
    tensors = {'w.' + k: v.contiguous() for k, v in t.items()}
    tensors['in.image'] = img
    tensors['out.tokens'] = out.contiguous()
    save_file(tensors, os.path.join(HERE, 'g4_clip.safetensors'))
    json.dump(dict(patch=P, dim=D, depth=L, heads=NH, mlp_ratio=4, ln_eps=1e-5, pre_norm=True, img_size=[S, S],
                   in_chans=3), open(os.path.join(HERE, 'g4_clip.json'), 'w'))
    print('g4_clip', tuple(out.shape))


def gen_swin(tag, H, W, depths, heads, window, C0=16):
    import transformers
    cfg = transformers.SwinConfig(image_size=(H, W), patch_size=4, num_channels=3, embed_dim=C0, depths=list(depths),
                                  num_heads=list(heads), window_size=window, mlp_ratio=4.0, qkv_bias=True,
                                  hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, drop_path_rate=0.0,
                                  hidden_act='gelu', layer_norm_eps=1e-5)
    m = transformers.SwinModel(cfg, add_pooling_layer=False).eval()
    randomise(m, 13)
    img = torch.randn(2, 3, H, W, generator=torch.Generator().manual_seed(5))
    with torch.no_grad():
        out = m(pixel_values=img).last_hidden_state
    sd = m.state_dict()
    t = {'patch_embed.proj.weight': sd['embeddings.patch_embeddings.projection.weight'],
         'patch_embed.proj.bias': sd['embeddings.patch_embeddings.projection.bias'],
         'patch_embed.norm.weight': sd['embeddings.norm.weight'], 'patch_embed.norm.bias': sd['embeddings.norm.bias'],
         'norm.weight': sd['layernorm.weight'], 'norm.bias': sd['layernorm.bias']}
    for si, depth in enumerate(depths):
        if si > 0:  # HF downsamples at the END of stage si-1; timm at the START of stage si
            for n in ('norm.weight', 'norm.bias', 'reduction.weight'):
                t[f'layers.{si}.downsample.{n}'] = sd[f'encoder.layers.{si - 1}.downsample.{n}']
        for bi in range(depth):
            hp, tp = f'encoder.layers.{si}.blocks.{bi}.', f'layers.{si}.blocks.{bi}.'
            t[tp + 'attn.relative_position_bias_table'] = sd[hp + 'attention.relative_position_bias.relative_position_bias_table']
            for wb in ('weight', 'bias'):
                t[tp + 'norm1.' + wb] = sd[hp + 'layernorm_before.' + wb]
                t[tp + 'norm2.' + wb] = sd[hp + 'layernorm_after.' + wb]
                t[tp + 'attn.qkv.' + wb] = torch.cat([sd[hp + f'attention.{n}_proj.' + wb] for n in ('q', 'k', 'v')], 0)
                t[tp + 'attn.proj.' + wb] = sd[hp + 'attention.o_proj.' + wb]
                t[tp + 'mlp.fc1.' + wb] = sd[hp + 'mlp.fc1.' + wb]
                t[tp + 'mlp.fc2.' + wb] = sd[hp + 'mlp.fc2.' + wb]
    tensors = {'w.' + k: v.contiguous() for k, v in t.items()}
    tensors['in.image'] = img
    tensors['out.tokens'] = out.contiguous()
    save_file(tensors, os.path.join(HERE, f'{tag}.safetensors'))
    json.dump(dict(patch=4, embed_dim=C0, depths=list(depths), heads=list(heads), window=window, mlp_ratio=4,
                   ln_eps=1e-5, img_size=[H, W], in_chans=3), open(os.path.join(HERE, f'{tag}.json'), 'w'))
    print(tag, tuple(out.shape))


# ------------------------------------------------------------------ G5: optimiser / schedule / clip
def gen_optim():
    gen = torch.Generator().manual_seed(21)
    shapes = [(5, 7), (11,), (3, 4, 2)]
    params = [torch.nn.Parameter(torch.randn(s, generator=gen)) for s in shapes]
    opt = torch.optim.AdamW(params, lr=3e-4, betas=(0.9, 0.98), eps=1e-6, weight_decay=0.0)
    tensors = {}
    for i, p in enumerate(params):
        tensors[f'p0.{i}'] = p.detach().clone()
    norms = []
    for step in range(3):
        for i, p in enumerate(params):
            p.grad = torch.randn(p.shape, generator=gen) * (3.0 if step == 1 else 0.1)
            tensors[f'g{step}.{i}'] = p.grad.clone()
        total = torch.nn.utils.clip_grad_norm_(params, 1.0)
        norms.append(float(total))
        for i, p in enumerate(params):
            tensors[f'gclip{step}.{i}'] = p.grad.clone()
        opt.step()
        for i, p in enumerate(params):
            tensors[f'p{step + 1}.{i}'] = p.detach().clone()
    save_file(tensors, os.path.join(HERE, 'g5_optim.safetensors'))
    # cosine schedule known answers by closed form: base 5e-4, W=5 intervals x U=10, E=100 x U
    base, warm, tin = 5e-4, 50, 1000
    lrs = {}
    for t in (0, 1, 25, 49, 50, 51, 500, 999, 1000, 1200):
        if t < warm:
            lr = t * base / warm
        elif t < tin:
            lr = 0.5 * base * (1 + math.cos(math.pi * t / tin))
        else:
            lr = 0.0
        lrs[str(t)] = lr
    json.dump(dict(lr=3e-4, betas=[0.9, 0.98], eps=1e-6, clip=1.0, grad_norms=norms, n_tensors=len(shapes),
                   sched=dict(base=base, warmup_t=warm, t_initial=tin, lrs=lrs)),
              open(os.path.join(HERE, 'g5_optim.json'), 'w'), indent=1)
    print('g5 norms', norms)


if __name__ == '__main__':
    gen_decoder('g1_decoder_tiny', d_model=64, heads=4, ffn=128, layers=2, vocab=515, T=15, S=10, B=2)
    gen_decoder('g2_decoder_hd64', d_model=256, heads=4, ffn=1024, layers=2, vocab=1027, T=127, S=49, B=2,
                logit_cols=16, all_grads=False)
    gen_preprocess()
    gen_vit()
    gen_clip()
    gen_swin('g4_swin_shift', 64, 64, (2, 2), (2, 4), 4)       # 16x16 -> 8x8, shifted windows in both stages
    gen_swin('g4_swin_clamp', 64, 32, (2, 2), (2, 4), 4)       # 16x8 -> 8x4: window == min side -> shift 0
    gen_optim()
    gen_finetune()
