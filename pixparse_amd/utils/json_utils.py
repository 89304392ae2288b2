"""Donut-style JSON <-> token-sequence conversion used by the fine-tune collators and the eval tasks
(behaviour of the reference's utils/json_utils.py:11-64 `json2token` and :67-116 `token2json`).

  {"menu": [{"nm": "latte", "cnt": "2"}, {"nm": "tea"}], "total": {"price": "9"}}
    -> <s_menu><s_nm>latte</s_nm><s_cnt>2</s_cnt><sep/><s_nm>tea</s_nm></s_menu><s_total><s_price>9</s_price></s_total>

Return-shape quirks are kept because callers unpack them: a dict / list / leaf returns ``(text, special_tokens)``,
the ``{"text_sequence": s}`` shortcut returns the bare string. The reference's mutable default list (which leaks
key tokens from one call into the next) is NOT reproduced: it never changes the text, only grows the second value."""
import re
from typing import Any, List, Optional, Tuple, Union


def json2token(obj: Any, tokenizer_all_special_tokens: List[str], additional_special_tokens: Optional[List[str]] = None,
               update_special_tokens_for_json_key: bool = True, sort_json_key: bool = True) -> Union[str, Tuple[str, List[str]]]:
    extra = [] if additional_special_tokens is None else additional_special_tokens
    if isinstance(obj, dict):
        if len(obj) == 1 and 'text_sequence' in obj:
            return obj['text_sequence']
        keys = sorted(obj.keys(), reverse=True) if sort_json_key else list(obj.keys())
        text = ''
        for k in keys:
            if update_special_tokens_for_json_key:
                extra.extend([f'<s_{k}>', f'</s_{k}>'])
            inner = json2token(obj[k], tokenizer_all_special_tokens, extra, update_special_tokens_for_json_key, sort_json_key)
            if isinstance(inner, tuple):
                inner, extra = inner
            text += f'<s_{k}>' + inner + f'</s_{k}>'
        return text, list(set(extra))
    if isinstance(obj, list):
        parts = []
        for item in obj:
            inner = json2token(item, tokenizer_all_special_tokens, extra, update_special_tokens_for_json_key, sort_json_key)
            if isinstance(inner, tuple):
                inner, extra = inner
            parts.append(inner)
        return '<sep/>'.join(parts), list(set(extra))
    leaf = str(obj)
    if f'<{leaf}/>' in tokenizer_all_special_tokens or f'<{leaf}/>' in extra:
        leaf = f'<{leaf}/>'   # categorical special token
    return leaf, list(set(extra))


def token2json(tokens: str, added_vocab=None, is_inner_value: bool = False):
    """inverse of json2token on generated text; unmatched start tags are dropped, leaves split on <sep/>"""
    added_vocab = {} if added_vocab is None else added_vocab
    output = {}
    while tokens:
        start = re.search(r'<s_(.*?)>', tokens, re.IGNORECASE)
        if start is None:
            break
        key = start.group(1)
        end = re.search(rf'</s_{key}>', tokens, re.IGNORECASE)
        start_tok = start.group()
        if end is None:
            tokens = tokens.replace(start_tok, '')
            continue
        end_tok = end.group()
        content = re.search(f'{re.escape(start_tok)}(.*?){re.escape(end_tok)}', tokens, re.IGNORECASE)
        if content is not None:
            body = content.group(1).strip()
            if '<s_' in body and '</s_' in body:   # non-leaf
                value = token2json(body, added_vocab, True)
                if value:
                    output[key] = value[0] if len(value) == 1 else value
            else:
                leaves = []
                for leaf in body.split('<sep/>'):
                    leaf = leaf.strip()
                    if leaf in added_vocab and leaf[0] == '<' and leaf[-2:] == '/>':
                        leaf = leaf[1:-2]
                    leaves.append(leaf)
                output[key] = leaves[0] if len(leaves) == 1 else leaves
        tokens = tokens[tokens.find(end_tok) + len(end_tok):].strip()
        if tokens[:6] == '<sep/>':
            return [output] + token2json(tokens[6:], added_vocab, True)
    if len(output):
        return [output] if is_inner_value else output
    return [] if is_inner_value else {'text_sequence': tokens}
