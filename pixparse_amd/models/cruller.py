"""Cruller = image encoder + BART text decoder (ref: models/cruller.py:8-21), MI355X-native.

Same surface as the reference module -- ``Cruller(cfg)``, ``.image_encoder.trunk`` (with
``pretrained_cfg``), ``.text_decoder.trunk.resize_token_embeddings(n)``,
``forward(image_input, text_input)['logits']``, ``state_dict()`` with the timm / HF key names --
but the parameters are views into one flat fp32 arena and all arithmetic is the explicit
forward/backward engine over libcruller_hip.so (layers/engines.py).  There is no CPU path:
calling forward on a CPU-resident model raises.
"""
from collections import OrderedDict
from typing import Callable, Optional

import torch
import torch.nn as nn

from .. import ops
from ..layers.arena import ParamArena
from ..layers.engines import BartEngine, Buffers, SwinEngine, ViTEngine
from .archs import BART_ARCHS, SWIN_ARCHS, VIT_ARCHS
from .config import ModelCfg

ENC_PREFIX = 'image_encoder.trunk.'
DEC_PREFIX = 'text_decoder.trunk.'



import os as _os
_WGRAD_OVERWRITE = _os.environ.get('PIXPARSE_AMD_WGRAD_OVERWRITE', '1') != '0'      # A/B switch (INTEGRATION.md knob table)

class _Container(nn.Module):
    """bare module used to reproduce the reference's state_dict key hierarchy"""


class _DecoderTrunk(_Container):
    def __init__(self, owner):
        super().__init__()
        object.__setattr__(self, '_owner', owner)

    def resize_token_embeddings(self, new_num_tokens: int):
        self._owner._resize_vocab(int(new_num_tokens))
        return self.model.decoder.embed_tokens


class _TextDecoder(_Container):
    def __init__(self, owner):
        super().__init__()
        object.__setattr__(self, '_owner', owner)

    def prepare_inputs_for_inference(self, input_ids, encoder_outputs, pad_token_id, past_key_values=None, past=None,
                                     use_cache=None, attention_mask=None):
        """ref: models/text_decoder_hf.py:47-78"""
        if past is not None:
            past_key_values = past
        attention_mask = input_ids.ne(pad_token_id).long()
        if past_key_values is not None:
            input_ids = input_ids[:, -1:]
        return {'input_ids': input_ids, 'attention_mask': attention_mask, 'past_key_values': past_key_values,
                'use_cache': use_cache, 'encoder_hidden_states': encoder_outputs}


    def forward(self, input_ids, encoder_hidden_states, attention_mask=None, past_key_values=None, use_cache=None, **_):
        """ref models/text_decoder_hf.py:39-45: the full (uncached) decoder over input_ids [B, T] -> .logits [B, T, V].
        Generation proper goes through Cruller.decode_begin / decode_step (KV cache); this is the drop-in for callers
        that re-run the decoder on the whole prefix like utils/ocr_utils.py:181-187."""
        owner = self._owner
        _, dec, _ = owner._ensure_engines()
        dec.drop = None       # never the mask of an earlier training step (only forward_loss() sets one, backward() clears it)
        B, S, D = encoder_hidden_states.shape
        enc16 = encoder_hidden_states.reshape(B * S, D).to(torch.bfloat16).contiguous()
        T = input_ids.shape[1]
        logits = dec.forward(input_ids.contiguous(), enc16, S)
        return CausalLMOutput(logits=logits.view(B, T, dec.Vp)[:, :, :owner.vocab_size], past_key_values=None)


class _ImageEncoder(_Container):
    def __init__(self, owner):
        super().__init__()
        object.__setattr__(self, '_owner', owner)

    def forward(self, image_input):
        """ref models/image_encoder_timm.py forward: features [B, S, D]"""
        return self._owner.encode_image(image_input)


class CausalLMOutput(dict):
    """mapping with attribute access, like transformers' ModelOutput (train_step indexes ['logits'],
    utils/ocr_utils.py reads .logits)"""
    __getattr__ = dict.get


def _set_param(root: nn.Module, dotted: str, p: nn.Parameter, factories=None):
    parts = dotted.split('.')
    mod = root
    for i, name in enumerate(parts[:-1]):
        nxt = mod._modules.get(name)
        if nxt is None:
            nxt = _Container()
            mod.add_module(name, nxt)
        mod = nxt
    mod.register_parameter(parts[-1], p)


HEAD_PAD = 32   # classifier rows are padded to 32: the head's dgrad contracts over the class dimension (NN GEMM: K % 32 == 0)


class Cruller(nn.Module):
    def __init__(self, cfg: ModelCfg, vocab_size: Optional[int] = None):
        super().__init__()
        self.cfg = cfg
        ie, td = cfg.image_encoder, cfg.text_decoder
        assert ie.name, 'image encoder name required'            # image_encoder_timm.py:8
        assert ie.image_fmt in ('L', 'RGB')                       # image_encoder_timm.py:12
        assert td.name, 'text decoder name required'              # text_decoder_hf.py:11
        if ie.name in SWIN_ARCHS:
            self.enc_kind, self.enc_arch = 'swin', SWIN_ARCHS[ie.name]
        elif ie.name in VIT_ARCHS:
            self.enc_kind, self.enc_arch = 'vit', VIT_ARCHS[ie.name]
        else:
            raise ValueError(f'unknown image encoder {ie.name!r}; known: {sorted(VIT_ARCHS) + sorted(SWIN_ARCHS)}')
        if td.name not in BART_ARCHS:
            raise ValueError(f'unknown text decoder {td.name!r}; known: {sorted(BART_ARCHS)}')
        self.dec_arch = BART_ARCHS[td.name]
        self.in_chans = 1 if ie.image_fmt == 'L' else 3
        self.img_size = tuple(ie.image_size) if ie.image_size is not None else (224, 224)
        self.n_layers = td.num_decoder_layers if td.num_decoder_layers is not None else 6
        self.max_length = td.max_length if td.max_length is not None else 1024
        self.vocab_size = int(vocab_size or self.dec_arch['vocab'])
        enc_dim = self.enc_arch['dim'] if self.enc_kind == 'vit' else self.enc_arch['embed_dim'] * 2 ** (len(self.enc_arch['depths']) - 1)
        assert enc_dim == self.dec_arch['d_model'], f'encoder width {enc_dim} != decoder d_model {self.dec_arch["d_model"]}'
        self.arena: Optional[ParamArena] = None
        self._engines = None
        self._head = None            # (num_classes, in_features) once add_classifier_head() ran (cruller_finetune_xent)
        self._build()
        self.reset_parameters()
        # pretrained=True (the reference default, image_encoder_timm.py:13-20 / text_decoder_hf.py:25-31): local weights when
        # they can be found, otherwise a loud warning -- never a silent random init
        from .pretrained import load_pretrained
        self.pretrained_sources = load_pretrained(self)

    # ------------------------------------------------------------------ structure
    def _enc_shapes(self):
        cls = SwinEngine if self.enc_kind == 'swin' else ViTEngine
        return cls.param_shapes(self.enc_arch, self.in_chans, self.img_size)

    def _build(self):
        arena = ParamArena()
        for item in self._enc_shapes():
            arena.add(ENC_PREFIX + item[0], item[1])
        for item in BartEngine.param_shapes(self.dec_arch, self.n_layers, self.vocab_size, self.max_length):
            arena.add(DEC_PREFIX + item[0], item[1], item[2] if len(item) > 2 else None)
        if self._head is not None:      # LAST in the arena: its gradients are the first the backward sweep completes
            nc, feat = self._head
            arena.add('final_fc.weight', (nc, feat), HEAD_PAD * feat)
            arena.add('final_fc.bias', (nc,), HEAD_PAD)
        arena.materialize('cpu')
        self.arena = arena
        self._engines = None
        # module tree mirroring the reference checkpoint keys
        self._modules.pop('final_fc', None)
        self._modules.pop('image_encoder', None)
        self._modules.pop('text_decoder', None)
        self.image_encoder = _ImageEncoder(self)
        self.image_encoder.trunk = _Container()
        self.image_encoder.trunk.pretrained_cfg = {'mean': self.enc_arch['mean'], 'std': self.enc_arch['std']}
        self.text_decoder = _TextDecoder(self)
        self.text_decoder.trunk = _DecoderTrunk(self)
        self._pmap = OrderedDict()
        for name in arena.entries:
            p = nn.Parameter(arena.param(name), requires_grad=True)
            self._pmap[name] = p
            _set_param(self, name, p)
        # tied LM head (hf:1224-1234): same Parameter object under both keys
        self.text_decoder.trunk.add_module('lm_head', _Container())
        self.text_decoder.trunk.lm_head.register_parameter('weight', self._pmap[DEC_PREFIX + 'model.decoder.embed_tokens.weight'])

    def reset_parameters(self, std: float = 0.02):
        """timm / HF style random init (pretrained weights cannot be fetched offline): N(0, .02) matrices and
        embeddings, zero biases, unit LayerNorm scales, zero row for the pad embedding (id 1)."""
        with torch.no_grad():
            for name, p in self._pmap.items():
                if name.endswith('.bias'):
                    p.zero_()
                elif 'norm' in name.rsplit('.', 2)[-2] and name.endswith('.weight'):
                    p.fill_(1.0)
                elif name.endswith('cls_token'):
                    p.normal_(0.0, 1e-6)
                else:
                    p.normal_(0.0, std)
            self._pmap[DEC_PREFIX + 'model.decoder.embed_tokens.weight'][1].zero_()

    def _resize_vocab(self, n: int):
        if n == self.vocab_size:
            return
        assert self.arena.p.device.type == 'cpu' and self.arena.g is None, 'resize_token_embeddings must precede train_setup()'
        old = {k: v.detach().clone() for k, v in self._pmap.items()}
        old_v = self.vocab_size
        self.vocab_size = n
        self._build()
        key = DEC_PREFIX + 'model.decoder.embed_tokens.weight'
        with torch.no_grad():
            for k, p in self._pmap.items():
                if k != key:
                    p.copy_(old[k])
            keep = min(old_v, n)
            self._pmap[key][:keep].copy_(old[key][:keep])
            if n > old_v:  # transformers 5 draws new rows around the old embeddings' mean; use the mean + small noise
                mean = old[key].mean(0, keepdim=True)
                self._pmap[key][old_v:].copy_(mean + 1e-3 * old[key].std() * torch.randn(n - old_v, mean.shape[1]))

    def add_classifier_head(self, num_classes: int, in_features: Optional[int] = None, seed: Optional[int] = None):
        """the classification fine-tune of the reference (task_cruller_finetune_xent.py:143-150): image encoder -> token 0 (GetCLSToken) ->
        nn.Linear(in_features, num_classes).  The head joins the parameter arena behind the decoder (nn.Linear's default init), so the same
        clip-norm / AdamW / bucketed all-reduce see it; classify_loss() / classify_backward() run encoder + head only (the decoder's
        gradients stay zero: with weight_decay 0, AdamW leaves it untouched, like the reference's optimiser that does not own it)."""
        if self.enc_kind != 'vit':
            raise NotImplementedError('the reference takes token 0 of the encoder output (GetCLSToken: x[:, 0, :]), which is a class token '
                                      'only for the ViT encoders; a Swin feature map [B, H, W, C] does not pass its CrossEntropyLoss either')
        feat = self.enc_arch['dim']
        if in_features is not None and in_features != feat:
            raise ValueError(f'classifier head expects {in_features} features, the encoder produces {feat}')
        assert 0 < num_classes <= HEAD_PAD
        assert self.arena.p.device.type == 'cpu' and self.arena.g is None, 'add_classifier_head must precede train_setup()'
        old = {k: v.detach().clone() for k, v in self._pmap.items()}
        self._head = (int(num_classes), int(feat))
        self._build()
        gen = torch.Generator().manual_seed(seed) if seed is not None else None
        bound = feat ** -0.5                  # nn.Linear.reset_parameters: kaiming_uniform(a = sqrt 5) = U(-1/sqrt(in), 1/sqrt(in)) for both
        with torch.no_grad():
            for k, p in self._pmap.items():
                if k in old:
                    p.copy_(old[k])
                else:
                    p.copy_((torch.rand(p.shape, generator=gen) * 2 - 1) * bound)
        return self

    # ------------------------------------------------------------------ device placement
    def _apply(self, fn, recurse=True):
        probe = fn(torch.empty(0, dtype=torch.float32, device=self.arena.p.device))
        if probe.dtype != torch.float32:
            raise TypeError('Cruller keeps fp32 master parameters; bf16 compute is internal (autocast policy)')
        self.arena.apply_(fn)
        for name, p in self._pmap.items():
            p.data = self.arena.param(name)
            if p.grad is not None:
                p.grad = None
        self._engines = None
        return self

    @property
    def device(self):
        return self.arena.p.device

    # ------------------------------------------------------------------ engines
    def _ensure_engines(self):
        if self._engines is not None:
            return self._engines
        if self.device.type != 'cuda':
            raise RuntimeError('Cruller runs only on an MI355X (HIP): move the model to a cuda device; there is no CPU path')
        from .. import hip
        hip.load()
        self.arena.alloc_shadow()
        bufs = Buffers(self.device)
        cls = SwinEngine if self.enc_kind == 'swin' else ViTEngine
        enc = cls(self.enc_arch, self.in_chans, self.img_size, self.arena, ENC_PREFIX, bufs)
        dec = BartEngine(self.dec_arch, self.n_layers, self.vocab_size, self.max_length, self.arena, DEC_PREFIX, bufs)
        self._engines = (enc, dec, bufs)
        self.refresh_shadows()
        return self._engines

    def refresh_shadows(self, full: bool = True):
        """bf16 copies of the weights (what the GEMMs read). ``full`` re-casts the whole arena (after loading a
        checkpoint / init); the fused AdamW kernel keeps it current afterwards, only the padded conv operand
        is re-made each step."""
        enc, dec, _ = self._engines
        if full:
            ops.cast_bf16(self.arena.p, self.arena.pb)
        enc.refresh_shadows()

    # ------------------------------------------------------------------ compute
    def encode(self, image_input: torch.Tensor):
        enc, _, _ = self._ensure_engines()
        assert image_input.shape[1:] == (self.in_chans, *self.img_size), \
            f'image {tuple(image_input.shape)} does not match the configured {self.in_chans}x{self.img_size}'  # timm strict img size
        enc32, enc16 = enc.forward(image_input.contiguous().float())
        return enc32, enc16

    def set_train_dropout(self, enabled: bool, seed: int = 0):
        """Opt-in hidden-state dropout of the text decoder inside forward_loss() / backward(): the reference's
        decoder runs it only when built with pretrained=False (BartForCausalLM from_config stays in train mode, p = config.dropout;
        text_decoder_hf.py:25-33); parity runs, bench.py and every eval / generation path keep it off.  The mask of micro-step s is
        a pure function of (seed, s, site, element) -- crl_dropout in include/crl.h.  Covers all four train-mode regularisers of the
        reference's models: hidden-state dropout, attention-probability and activation dropout (bart-base) and the Swin encoder's
        drop-path (timm's default drop_path_rate 0.1; create_model leaves the encoder in train mode)."""
        p = float(self.dec_arch.get('dropout', 0.0))
        p_attn = float(self.dec_arch.get('attention_dropout', 0.0))       # bart-base: dropout of the attention probabilities (hf BartAttention)
        p_act = float(self.dec_arch.get('activation_dropout', 0.0))       # bart-base: dropout behind the FFN's GELU (hf:384)
        p_path = float(self.enc_arch.get('drop_path', 0.0)) if self.enc_kind == 'swin' else 0.0   # timm Swin drop-path (train-mode encoder)
        self._drop = (p, int(seed), p_attn, p_act, p_path) if enabled and max(p, p_attn, p_act, p_path) > 0 else None
        self._drop_step = 0

    def forward(self, image_input: torch.Tensor, text_input: torch.Tensor, _drop=None):
        """ref models/cruller.py:14-21 -> output['logits'] bf16 [B, T, V] (a view of the padded logits buffer)."""
        enc, dec, _ = self._ensure_engines()
        dec.drop = _drop
        if self.enc_kind == 'swin':
            enc.drop = _drop          # drop-path of the encoder's residual branches (None outside forward_loss / backward pairs)
        enc32, enc16 = self.encode(image_input)
        B, T = text_input.shape
        logits = dec.forward(text_input.contiguous(), enc16, enc.out_tokens())
        return CausalLMOutput(logits=logits.view(B, T, dec.Vp)[:, :, :self.vocab_size],
                              encoder_last_hidden_state=enc32.view(B, enc.out_tokens(), -1))

    # ------------------------------------------------------------------ generation
    def encode_image(self, image_input: torch.Tensor) -> torch.Tensor:
        """what the reference's eval tasks call `model.image_encoder(image)`: last hidden states fp32 [B, S, D]"""
        enc, _, _ = self._ensure_engines()
        if self.enc_kind == 'swin':
            enc.drop = None           # eval path: never the drop-path masks of an earlier training step
        enc32, _ = self.encode(image_input)
        return enc32.view(image_input.shape[0], enc.out_tokens(), -1)

    def decode_begin(self, encoder_outputs: torch.Tensor, max_len: int):
        """encoder_outputs [B, S, D] (fp32 or bf16): builds the cross-attention K/V and empty self-attention caches"""
        _, dec, _ = self._ensure_engines()
        dec.drop = None
        B, S, D = encoder_outputs.shape
        enc16 = encoder_outputs.reshape(B * S, D).to(torch.bfloat16).contiguous()
        dec.decode_begin(enc16, B, S, int(max_len))

    def decode_prefill(self, prompt_ids: torch.Tensor) -> None:
        """prompt_ids [B, P]: the first P tokens of every sequence in ONE decoder pass that fills cache rows 0..P-1 (no logits);
        feed the prompt's last token through decode_step() next"""
        _, dec, _ = self._ensure_engines()
        dec.decode_prefill(prompt_ids.contiguous())

    def decode_step(self, input_ids: torch.Tensor) -> torch.Tensor:
        """input_ids [B, 1]: the next token of every sequence -> next-token logits bf16 [B, V]"""
        _, dec, _ = self._ensure_engines()
        return dec.decode_step(input_ids.contiguous())[:, :self.vocab_size]

    def generate_greedy(self, encoder_outputs: torch.Tensor, prompt_id, eos_id: int, max_steps: int,
                        use_graph: bool = True, return_logits: bool = False, check_every: int = 8):
        """The reference's greedy loop (utils/ocr_utils.py:165-197) on the KV-cache decode path: every sequence starts
        from prompt_id; iteration r appends arg-max token r unless ALL sequences have produced eos by then (the token of
        that iteration is not appended). prompt_id may also be a multi-token prompt (a list of ids, the same for every sequence, or an
        int64 tensor [B, P]): all but its last token go through one prefill pass (decode_prefill), the last one starts the loop;
        the returned ids then begin with the whole prompt. The step has no host-visible state (position, cache row and key count come from
        a device counter), so after one eager iteration it is captured in a hipGraph and replayed; the host only reads
        the "all finished at iteration" flag every `check_every` replays. Returns ids [B, n] (and the per-iteration
        fp32 logits when return_logits, which forces the eager path)."""
        _, dec, bufs = self._ensure_engines()
        dev = self.device
        B = encoder_outputs.shape[0]
        V = self.vocab_size
        if isinstance(prompt_id, int):
            prompt = torch.full((B, 1), prompt_id, dtype=torch.int64, device=dev)
        else:
            prompt = torch.as_tensor(prompt_id, dtype=torch.int64, device=dev)
            prompt = prompt.view(1, -1).expand(B, -1).contiguous() if prompt.dim() == 1 else prompt
        P_ = prompt.shape[1]
        self.decode_begin(encoder_outputs, max_steps + P_)
        if P_ > 1:
            self.decode_prefill(prompt[:, :-1])
        ids = bufs.get('gen.ids', (B, 1), torch.int64)
        ids.copy_(prompt[:, -1:])
        tokens = bufs.get('gen.tokens', (B, max_steps + 1), torch.int64)
        tokens.copy_(prompt[:, -1:].expand(B, max_steps + 1))
        finished = bufs.get('gen.finished', (B,), torch.bool)
        finished.zero_()
        done_at = bufs.get('gen.done_at', (1,), torch.int32)
        done_at.fill_(-1)
        r = bufs.get('gen.r', (1,), torch.int32)
        r.zero_()
        steps = []

        def iteration():
            logits = dec.decode_step(ids)[:, :V]
            if return_logits:
                steps.append(logits.float().clone())
            nxt = torch.argmax(logits, dim=-1, keepdim=True)
            finished.logical_or_(nxt[:, 0] == eos_id)
            first = finished.all() & (done_at < 0)
            done_at.copy_(torch.where(first, r, done_at))
            tokens.scatter_(1, (r + 1).to(torch.int64).view(1, 1).expand(B, 1), nxt)
            ids.copy_(nxt)
            r.add_(1)

        n_done = 0
        if max_steps > 0:
            iteration()
            n_done = 1
        graph = None
        if use_graph and not return_logits and max_steps > 1:
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                iteration()
            dec.gen['t'] -= 1       # capture ran the Python body once without executing a step
        while n_done < max_steps:
            if int(done_at.item()) >= 0:
                break
            for _ in range(min(check_every, max_steps - n_done)):
                if graph is not None:
                    graph.replay()
                    dec.gen['t'] += 1
                else:
                    iteration()
                n_done += 1
        d = int(done_at.item())
        n_tokens = 1 + (d if d >= 0 else n_done)
        out = torch.cat([prompt[:, :-1], tokens[:, :n_tokens]], dim=1)
        return (out, steps[:n_tokens if d < 0 else d + 1]) if return_logits else out

    def forward_loss(self, image_input, text_input, text_target, loss_mul: float = 1.0, grad_mul: float = 1.0,
                     grad_mul_dev: Optional[torch.Tensor] = None):
        """forward + shifted-token cross-entropy; leaves d(loss*grad_mul)/dlogits in the logits buffer.
        returns the device scalar loss (fp32, = mean NLL * loss_mul)."""
        enc, dec, bufs = self._ensure_engines()
        drop = None
        if getattr(self, '_drop', None) is not None:
            drop = ops.DropSpec(self._drop[0], self._drop[1], self._drop_step, *self._drop[2:])
            self._drop_step += 1
        self.forward(image_input, text_input, _drop=drop)   # dec.drop stays set for the matching backward()
        B, T = text_input.shape
        M = B * T
        logits = bufs.t['dec.logits']
        self._loss = bufs.get('loss', (1,), torch.float32)
        n_valid = bufs.get('n_valid', (1,), torch.int32)
        row_loss = bufs.get('row_loss', (M,), torch.float32)
        self._target = text_target.contiguous().view(-1)
        ops.cross_entropy(logits, self._target, self.vocab_size, loss_mul, grad_mul, self._loss, n_valid, row_loss, logits, grad_mul_dev)
        return self._loss

    def backward(self, on_ready: Optional[Callable[[str], None]] = None, first_micro: bool = False):
        """backward of the last forward_loss(); weight gradients accumulate into arena.g.  ``on_ready(name)`` is
        called as the sweep passes arena entry ``name``: every gradient at or after it (layout order) is final.
        first_micro: the gradient arena holds zeros (the first micro-step behind an optimiser step that zero-filled it): the weight-gradient
        GEMMs then overwrite instead of adding to those zeros (layers/engines.py _Base.first_micro; same bits, 2.1 GB less traffic at cfg-3)."""
        enc, dec, bufs = self._ensure_engines()
        enc.first_micro = dec.first_micro = bool(first_micro) and _WGRAD_OVERWRITE
        assert self.arena.g is not None, 'call alloc_training_state() (train_setup) before backward'
        S = enc.out_tokens()
        denc = bufs.get('denc', (dec.B * S, dec.D), torch.float32)     # written (not accumulated) by the decoder's last layer first
        dec.backward(bufs.t['dec.logits'], bufs.t[enc.tag + '.norm.y16'], denc, on_ready)
        enc.backward(denc, on_ready)
        dec.drop = None       # the mask belongs to this forward_loss() / backward() pair only
        if self.enc_kind == 'swin':
            enc.drop = None

    # ------------------------------------------------------------------ classification head (cruller_finetune_xent)
    def classify(self, image_input: torch.Tensor) -> torch.Tensor:
        """logits bf16 [B, HEAD_PAD] (columns >= num_classes are padding) = final_fc(encoder(image)[:, 0, :])"""
        assert self._head is not None, 'add_classifier_head() first'
        enc, _, bufs = self._ensure_engines()
        _, enc16 = self.encode(image_input)
        B, S, feat = image_input.shape[0], enc.out_tokens(), self._head[1]
        self._cls16 = enc16.view(B, S, feat)[:, 0, :]                    # strided rows: the GEMMs take a leading dimension
        wb = self.arena.shadow('final_fc.weight', padded=True).view(HEAD_PAD, feat)
        logits = bufs.get('head.logits', (B, HEAD_PAD), torch.bfloat16)
        ops.linear_fwd(self._cls16, wb, self.arena.param('final_fc.bias', padded=True), logits)
        return logits

    def classify_loss(self, image_input, label, loss_mul: float = 1.0, grad_mul: float = 1.0, grad_mul_dev: Optional[torch.Tensor] = None):
        """autocast(bf16){ final_fc(encoder(image)[:, 0]) } -> CrossEntropyLoss(ignore_index=-100) (ref task_cruller_finetune_xent.py:237-247);
        leaves d(loss * grad_mul) / dlogits in the logits buffer, returns the device scalar loss"""
        logits = self.classify(image_input)
        _, _, bufs = self._engines
        B = logits.shape[0]
        self._loss = bufs.get('loss', (1,), torch.float32)
        n_valid = bufs.get('n_valid', (1,), torch.int32)
        row_loss = bufs.get('head.row_loss', (B,), torch.float32)
        ops.cross_entropy(logits, label.contiguous().view(-1), self._head[0], loss_mul, grad_mul, self._loss, n_valid, row_loss, logits, grad_mul_dev)
        return self._loss

    def classify_backward(self, on_ready: Optional[Callable[[str], None]] = None):
        """backward of the last classify_loss(): head wgrad / bias grad / dgrad, the class-token rows of d(encoder output), encoder backward"""
        enc, _, bufs = self._ensure_engines()
        assert self.arena.g is not None, 'call alloc_training_state() (train_setup) before backward'
        nc, feat = self._head
        dlogits = bufs.t['head.logits']
        B, S = dlogits.shape[0], enc.out_tokens()
        ops.linear_wgrad(dlogits, self._cls16, self.arena.grad('final_fc.weight', padded=True).view(HEAD_PAD, feat), True)
        ops.colsum(dlogits, self.arena.grad('final_fc.bias', padded=True), True)
        if on_ready is not None:
            on_ready('final_fc.weight')
        dcls = bufs.get('head.dcls', (B, feat), torch.bfloat16)
        ops.linear_dgrad(dlogits, self.arena.shadow('final_fc.weight', padded=True).view(HEAD_PAD, feat), dcls)
        denc = bufs.get('denc', (B * S, feat), torch.float32)
        denc.zero_()                                                      # only token 0 of every image receives a gradient
        denc.view(B, S, feat)[:, 0, :].copy_(dcls)
        enc.backward(denc, on_ready)

    def activation_bytes(self) -> int:
        return 0 if self._engines is None else self._engines[2].bytes()
