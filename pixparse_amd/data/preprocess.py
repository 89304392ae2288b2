"""Annotation -> (text, target) token tensors: the input contract of train_step
(ref: data/preprocess.py:9-131).  Pure host-side int64 work, kept semantically identical to the
reference including its quirk that the prompt prefix length is the SUM of the indices at which the
prompt-end token occurs, plus one."""
import logging
from typing import Callable

import torch

_logger = logging.getLogger(__name__)


def mask_targets(text: torch.Tensor, pad_token_id: int, prompt_end_token_id: int, ignore_id: int = -100) -> torch.Tensor:
    target = text.clone()
    target[target == pad_token_id] = ignore_id                     # never predict padding
    prefix = int(torch.nonzero(target == prompt_end_token_id).sum()) + 1
    target[:prefix] = ignore_id                                    # never predict the prompt
    return target


def _tokenize(tokenizer: Callable, s: str, max_length: int) -> torch.Tensor:
    return tokenizer(s, add_special_tokens=False, return_tensors='pt', max_length=max_length, padding='max_length',
                     truncation=True).input_ids[0]


def preprocess_text_anno(anno, tokenizer: Callable, max_position_embeddings: int, task_start_token: str,
                         prompt_end_token: str, ignore_id: int = -100, generator=None):
    text = _tokenize(tokenizer, task_start_token + anno + tokenizer.eos_token, max_position_embeddings)
    target = mask_targets(text, tokenizer.pad_token_id, tokenizer.convert_tokens_to_ids(prompt_end_token), ignore_id)
    return dict(text=[text], target=[target])


def get_next_valid_page_index(current_index: int, num_pages: int, anno: dict, retries: int = 10):
    for _ in range(retries):
        current_index = (current_index + 1) % num_pages
        if anno['pages'][current_index]['text']:
            return current_index
    raise RuntimeError(f'No non-empty page found after {retries} attempts')


def preprocess_ocr_anno(anno, tokenizer: Callable, max_position_embeddings: int, task_start_token: str,
                        prompt_end_token: str, ignore_id: int = -100, generator=None):
    if isinstance(anno, list):
        _logger.warning('Old [id, {}] annotation form found, correcting...')
        anno = anno[1]
    num_pages = len(anno['pages'])
    if not num_pages:
        raise RuntimeError('Empty annotation. Skipping...')
    pad_id = tokenizer.pad_token_id
    prompt_end_id = tokenizer.convert_tokens_to_ids(prompt_end_token)
    current_index = generator.randint(0, num_pages - 1)
    if not anno['pages'][current_index]['text']:
        current_index = get_next_valid_page_index(current_index, num_pages, anno)
    page_indices, text_pages, target_pages = [], [], []
    n_wanted_pages = min(1, num_pages)
    orig_text = None
    while len(text_pages) < n_wanted_pages:
        page = anno['pages'][current_index]
        if not page['text']:
            raise RuntimeError('No text on page, skipping...')
        orig_text = '\n'.join(page['text'])
        text = _tokenize(tokenizer, task_start_token + orig_text + tokenizer.eos_token, max_position_embeddings)
        text_pages.append(text)
        target_pages.append(mask_targets(text, pad_id, prompt_end_id, ignore_id))
        page_indices.append(current_index)
        current_index = get_next_valid_page_index(current_index, num_pages, anno)
    return dict(text=text_pages, target=target_pages), dict(page_indices=page_indices, num_pages=num_pages, orig_text=orig_text)
