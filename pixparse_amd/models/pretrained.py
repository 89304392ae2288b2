"""Loading pretrained timm / transformers weights into the flat-arena Cruller (`pretrained=True` of ImageEncoderCfg /
TextDecoderCfg).

The reference gets them from the hub:
  * encoder  ``timm.create_model(name, pretrained=True, in_chans=1|3, num_classes=0, global_pool='', img_size=...)``
             (models/image_encoder_timm.py:13-20): the timm checkpoint, with timm's two load-time adaptations --
             ``adapt_input_conv`` (a 1-channel model sums the RGB patch-embed kernels over the input channel) and
             ``resample_abs_pos_embed`` (the position table is resized to the configured grid: bicubic, antialiased,
             the class-token row kept);
  * decoder  ``AutoModelForCausalLM.from_pretrained('facebook/bart-*', config)`` with ``decoder_layers = n``
             (models/text_decoder_hf.py:23-33): ``BartForCausalLM`` picks ``model.decoder.*`` out of the seq2seq
             checkpoint, i.e. the FIRST n decoder layers, the shared token embedding (tied to ``lm_head``) and the learned
             positions; the task later grows the embedding with ``resize_token_embeddings``.
There is no network here, so the files must be local: ``cfg.pretrained_path`` (a file, or a directory searched for
``<name>.safetensors|.pt|.pth|.bin`` with '/' in the name replaced by '--'), else the directory / file named by the
environment variable ``PIXPARSE_AMD_WEIGHTS``.  When ``pretrained=True`` and nothing is found the model is
random-initialised and says so LOUDLY (a warning per component; ``PIXPARSE_AMD_STRICT_PRETRAINED=1`` turns it into an
error) -- it is never silent.
"""
import logging
import os
import warnings
from typing import Dict, Optional

import torch

_logger = logging.getLogger(__name__)
ENV_DIR, ENV_STRICT = 'PIXPARSE_AMD_WEIGHTS', 'PIXPARSE_AMD_STRICT_PRETRAINED'
_EXTS = ('.safetensors', '.pt', '.pth', '.bin')


class PretrainedWeightsMissing(UserWarning):
    pass


def find_weights(name: str, explicit: Optional[str] = None) -> Optional[str]:
    cands = [explicit, os.environ.get(ENV_DIR)]
    stem = name.replace('/', '--')
    for c in cands:
        if not c:
            continue
        if os.path.isfile(c):
            if c is explicit or os.path.splitext(os.path.basename(c))[0] == stem:
                return c
            continue
        if os.path.isdir(c):
            for ext in _EXTS:
                for fn in (stem + ext, os.path.join(stem, 'model' + ext), os.path.join(stem, 'pytorch_model' + ext)):
                    p = os.path.join(c, fn)
                    if os.path.isfile(p):
                        return p
    return None


def read_state(path: str) -> Dict[str, torch.Tensor]:
    if path.endswith('.safetensors'):
        from safetensors.torch import load_file
        sd = load_file(path)
    else:
        sd = torch.load(path, map_location='cpu', weights_only=True)
    for k in ('state_dict', 'model'):
        if isinstance(sd, dict) and k in sd and isinstance(sd[k], dict):
            sd = sd[k]
    return {k: v for k, v in sd.items() if torch.is_tensor(v)}


def missing(component: str, name: str) -> None:
    msg = (f'{component}: pretrained=True but no local weights for {name!r} (looked at cfg.pretrained_path and ${ENV_DIR}); '
           f'the reference would download them -- continuing with RANDOM INITIALISATION')
    if os.environ.get(ENV_STRICT, '0') == '1':
        raise FileNotFoundError(msg)
    warnings.warn(msg, PretrainedWeightsMissing, stacklevel=3)
    _logger.warning(msg)


# ------------------------------------------------------------------------------------------------ encoder (timm)
def adapt_input_conv(in_chans: int, w: torch.Tensor) -> torch.Tensor:
    """timm.models._manipulate.adapt_input_conv: [O, 3, kh, kw] -> [O, in_chans, kh, kw]"""
    w = w.float()
    O, I, kh, kw = w.shape
    if in_chans == I:
        return w
    if in_chans == 1:
        if I > 3:
            assert I % 3 == 0
            return w.reshape(O, I // 3, 3, kh, kw).sum(dim=2)
        return w.sum(dim=1, keepdim=True)
    if I != 3:
        raise NotImplementedError('weight format not supported by conversion')
    rep = -(-in_chans // 3)
    return w.repeat(1, rep, 1, 1)[:, :in_chans] * (3.0 / in_chans)


def resample_abs_pos_embed(pos: torch.Tensor, new_grid, num_prefix_tokens: int = 1) -> torch.Tensor:
    """timm.layers.resample_abs_pos_embed(interpolation='bicubic', antialias=True): [1, P + g*g, D] -> [1, P + gh*gw, D]"""
    gh, gw = new_grid
    n_new = gh * gw + num_prefix_tokens
    if pos.shape[1] == n_new and gh == gw:
        return pos
    prefix, grid = pos[:, :num_prefix_tokens], pos[:, num_prefix_tokens:]
    g_old = int(round(grid.shape[1] ** 0.5))
    assert g_old * g_old == grid.shape[1], 'pretrained position table is not a square grid'
    if (g_old, g_old) == (gh, gw):
        return pos
    D = grid.shape[-1]
    x = grid.float().reshape(1, g_old, g_old, D).permute(0, 3, 1, 2)
    x = torch.nn.functional.interpolate(x, size=(gh, gw), mode='bicubic', antialias=True, align_corners=False)
    x = x.permute(0, 2, 3, 1).reshape(1, gh * gw, D).to(pos.dtype)
    return torch.cat([prefix, x], dim=1)


def load_vit_weights(model, sd: Dict[str, torch.Tensor], prefix: str) -> int:
    """copy a timm VisionTransformer state dict into the arena parameters under `prefix`; returns the tensors copied"""
    a = model.enc_arch
    gh, gw = model.img_size[0] // a['patch'], model.img_size[1] // a['patch']
    n = 0
    with torch.no_grad():
        for name, p in model._pmap.items():
            if not name.startswith(prefix):
                continue
            k = name[len(prefix):]
            if k not in sd:
                raise KeyError(f'pretrained encoder checkpoint has no tensor {k!r}')
            v = sd[k]
            if k == 'patch_embed.proj.weight':
                v = adapt_input_conv(model.in_chans, v)
            elif k == 'pos_embed':
                v = resample_abs_pos_embed(v, (gh, gw), 1)
            if tuple(v.shape) != tuple(p.shape):
                raise ValueError(f'{k}: checkpoint shape {tuple(v.shape)} != model shape {tuple(p.shape)}')
            p.copy_(v.float())
            n += 1
    return n


def load_swin_weights(model, sd: Dict[str, torch.Tensor], prefix: str) -> int:
    n = 0
    with torch.no_grad():
        for name, p in model._pmap.items():
            if not name.startswith(prefix):
                continue
            k = name[len(prefix):]
            if k not in sd:
                raise KeyError(f'pretrained encoder checkpoint has no tensor {k!r}')
            v = sd[k]
            if k == 'patch_embed.proj.weight':
                v = adapt_input_conv(model.in_chans, v)
            if tuple(v.shape) != tuple(p.shape):   # the reference has the same limitation (image_encoder_timm.py:22-23 FIXME)
                raise ValueError(f'{k}: checkpoint shape {tuple(v.shape)} != model shape {tuple(p.shape)} '
                                 '(changing the Swin window / resolution of pretrained weights is not supported)')
            p.copy_(v.float())
            n += 1
    return n


# ------------------------------------------------------------------------------------------------ decoder (HF BART)
def load_bart_decoder_weights(model, sd: Dict[str, torch.Tensor], prefix: str) -> int:
    """BartForCausalLM.from_pretrained on a (seq2seq or causal-LM) BART checkpoint: `model.decoder.*` only, the first
    n layers, token embedding from `model.shared.weight` when the decoder copy is absent; rows are copied up to the
    smaller of the two vocabularies / position tables (transformers refuses a position-table mismatch; here the common
    rows are kept and the rest stays at its initialisation, with a warning)."""
    def pick(k):
        for cand in (k, k.replace('model.decoder.', 'decoder.'), 'model.' + k):
            if cand in sd:
                return sd[cand]
        if k.endswith('embed_tokens.weight'):
            for cand in ('model.shared.weight', 'shared.weight', 'lm_head.weight'):
                if cand in sd:
                    return sd[cand]
        return None
    n = 0
    with torch.no_grad():
        for name, p in model._pmap.items():
            if not name.startswith(prefix):
                continue
            k = name[len(prefix):]
            v = pick(k)
            if v is None:
                raise KeyError(f'pretrained decoder checkpoint has no tensor {k!r}')
            v = v.float()
            if tuple(v.shape) == tuple(p.shape):
                p.copy_(v)
            elif k.endswith(('embed_tokens.weight', 'embed_positions.weight')) and v.shape[1:] == p.shape[1:]:
                rows = min(v.shape[0], p.shape[0])
                p[:rows].copy_(v[:rows])
                if k.endswith('embed_positions.weight'):
                    _logger.warning(f'{k}: checkpoint has {v.shape[0]} positions, model {p.shape[0]}: first {rows} rows loaded')
            else:
                raise ValueError(f'{k}: checkpoint shape {tuple(v.shape)} != model shape {tuple(p.shape)}')
            n += 1
    return n


def load_pretrained(model) -> Dict[str, Optional[str]]:
    """apply cfg.image_encoder.pretrained / cfg.text_decoder.pretrained to a freshly initialised Cruller (CPU arena)"""
    from .cruller import DEC_PREFIX, ENC_PREFIX
    ie, td = model.cfg.image_encoder, model.cfg.text_decoder
    used = {'image_encoder': None, 'text_decoder': None}
    if ie.pretrained:
        path = find_weights(ie.name, getattr(ie, 'pretrained_path', None))
        if path is None:
            missing('image_encoder', ie.name)
        else:
            sd = read_state(path)
            (load_swin_weights if model.enc_kind == 'swin' else load_vit_weights)(model, sd, ENC_PREFIX)
            used['image_encoder'] = path
    if td.pretrained:
        path = find_weights(td.name, getattr(td, 'pretrained_path', None))
        if path is None:
            missing('text_decoder', td.name)
        else:
            load_bart_decoder_weights(model, read_state(path), DEC_PREFIX)
            used['text_decoder'] = path
    return used
