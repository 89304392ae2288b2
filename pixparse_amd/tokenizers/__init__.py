"""Tokenizer config + wrapper (ref: tokenizers/config.py:14-17, tokenizers/tokenizer_hf.py:6-18).

`TokenizerHF(cfg)` = `transformers.AutoTokenizer.from_pretrained(cfg.name)` like the reference.  This image has no
network and no cached `facebook/bart-large` files, so a byte-level stand-in with BART's special ids (bos 0, pad 1, eos 2,
vocab 50265) exists for synthetic benches and tests.  It is never substituted silently:
  * `TokenizerCfg(name='byte-bart')` selects it explicitly (bench.py and the tests do);
  * any other name is loaded through transformers from LOCAL files only (cache / directory); the hub is contacted only when
    PIXPARSE_AMD_ALLOW_HUB=1 says so -- the training boxes have no network, and every rank stalling in hub retries at start-up
    helps nobody; if that fails the error is logged at WARNING level with the reason and the stand-in is used -- its vocabulary is
    INCOMPATIBLE with real BART checkpoints -- unless PIXPARSE_AMD_STRICT_TOKENIZER=1, which re-raises."""
import logging
import os
import warnings
from dataclasses import dataclass

_logger = logging.getLogger(__name__)
BYTE_TOKENIZER = 'byte-bart'


@dataclass
class TokenizerCfg:
    name: str = 'facebook/bart-large'
    pretrained: bool = True


class ByteBartTokenizer:
    bos_token, pad_token, eos_token, unk_token = '<s>', '<pad>', '</s>', '<unk>'
    bos_token_id, pad_token_id, eos_token_id, unk_token_id = 0, 1, 2, 3

    def __init__(self, base_vocab: int = 50265):
        self.base_vocab = base_vocab
        self.added = {}

    def add_special_tokens(self, d):
        n = 0
        for t in d.get('additional_special_tokens', []):
            if t not in self.added:
                self.added[t] = self.base_vocab + len(self.added)
                n += 1
        return n

    def __len__(self):
        return self.base_vocab + len(self.added)

    @property
    def all_special_tokens(self):
        return [self.bos_token, self.eos_token, self.unk_token, self.pad_token, *self.added]

    def convert_tokens_to_ids(self, tok):
        fixed = {self.bos_token: 0, self.pad_token: 1, self.eos_token: 2, self.unk_token: 3}
        if tok in fixed:
            return fixed[tok]
        return self.added.get(tok, self.unk_token_id)

    def __call__(self, text, add_special_tokens=False, return_tensors='pt', max_length=None, padding='max_length', truncation=True):
        import torch
        specials = dict(self.added)
        specials.update({self.eos_token: 2, self.bos_token: 0, self.pad_token: 1})
        ids, i = [], 0
        while i < len(text):
            for s, sid in specials.items():
                if text.startswith(s, i):
                    ids.append(sid)
                    i += len(s)
                    break
            else:
                for b in text[i].encode('utf-8'):
                    ids.append(4 + b)
                i += 1
        if truncation and max_length is not None:
            ids = ids[:max_length]
        if padding == 'max_length' and max_length is not None:
            ids = ids + [self.pad_token_id] * (max_length - len(ids))

        class _Enc:
            pass
        e = _Enc()
        e.input_ids = torch.tensor([ids], dtype=torch.int64)
        return e

    def batch_decode(self, batch, skip_special_tokens=False):
        return [self.decode(ids, skip_special_tokens) for ids in (batch.tolist() if hasattr(batch, 'tolist') else batch)]

    def encode(self, text, add_special_tokens=False):
        return self(text, add_special_tokens=add_special_tokens, max_length=None, padding=False, truncation=False).input_ids[0].tolist()

    def decode(self, ids, skip_special_tokens=False):
        inv = {v: k for k, v in self.added.items()}
        out = bytearray()
        s = ''
        for t in ids:
            t = int(t)
            if 4 <= t < 260:
                out.append(t - 4)
            else:
                s += out.decode('utf-8', 'replace')
                out = bytearray()
                if not skip_special_tokens:
                    s += inv.get(t, {0: '<s>', 1: '<pad>', 2: '</s>'}.get(t, ''))
        return s + out.decode('utf-8', 'replace')


class TokenizerFallbackWarning(UserWarning):
    pass


class TokenizerHF:
    def __init__(self, cfg: TokenizerCfg):
        self.trunk = None
        if cfg.name == BYTE_TOKENIZER:
            self.trunk = ByteBartTokenizer()
            return
        try:
            import transformers
            try:
                self.trunk = transformers.AutoTokenizer.from_pretrained(cfg.name, local_files_only=True)
            except Exception:
                if os.environ.get('PIXPARSE_AMD_ALLOW_HUB', '0') != '1' or os.environ.get('HF_HUB_OFFLINE', '0') == '1':
                    raise
                self.trunk = transformers.AutoTokenizer.from_pretrained(cfg.name)
        except Exception as e:  # noqa: BLE001  (import error, cache miss, no network, typo in the name ...)
            if os.environ.get('PIXPARSE_AMD_STRICT_TOKENIZER', '0') == '1':
                raise
            msg = (f'tokenizer {cfg.name!r} could not be loaded ({type(e).__name__}: {str(e)[:200]}); using the byte-level stand-in '
                   f'({BYTE_TOKENIZER}): its vocabulary is NOT compatible with real BART checkpoints or real-data runs')
            warnings.warn(msg, TokenizerFallbackWarning, stacklevel=2)
            _logger.warning(msg)
            self.trunk = ByteBartTokenizer()
