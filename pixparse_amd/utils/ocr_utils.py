"""Greedy generation for the eval tasks / train-time OCR metric.
ref: utils/ocr_utils.py:143-197 (`generate_ocr`, `get_generated_tokens`) and :200-222 (`get_next_token`).

Same signatures and the same loop semantics as the reference -- every sequence starts from the prompt token, a sample
that has produced eos keeps being extended until ALL samples have, the token that completes the last sample is not
appended -- but where the reference re-runs the whole decoder on the growing prefix for every token, this goes through
`Cruller.decode_begin` / `decode_step`: cross-attention K/V projected once, self-attention K/V cached, one token of new
work per step (skinny HBM-bound projections + split-KV single-query attention), and the step -- free of host-visible
state -- is captured once in a hipGraph and replayed (`Cruller.generate_greedy`)."""
import re
from typing import List, Optional, Tuple

import torch


def get_next_token(next_token_logits: torch.Tensor, use_sample: bool = True, temperature: float = 5) -> Tuple[torch.Tensor, torch.Tensor]:
    """ref :200-222: sample from softmax(logits / temperature) or take the arg-max; returns ([B, 1] ids, probabilities)"""
    if use_sample:
        probs = torch.softmax(next_token_logits.float() / temperature, dim=-1)
        next_token_id = torch.multinomial(probs, num_samples=1)
    else:
        next_token_id = torch.argmax(next_token_logits, dim=-1, keepdim=True)
        probs = torch.ones_like(next_token_logits)
    return next_token_id, probs


def get_generated_tokens(model, tokenizer, encoder_outputs: torch.Tensor, device_env, max_recursion_length: int,
                         prompt_token: str, return_logits: bool = False, use_graph: bool = True):
    """ref :165-197. encoder_outputs [B, S, D] from `model.image_encoder(image)`; returns the token ids [B, n] (prompt
    token first). With return_logits also the list of per-step next-token logits (fp32 [B, V]) for parity checks."""
    prompt_id = tokenizer.trunk.encode(prompt_token, add_special_tokens=False)[0]
    return model.generate_greedy(encoder_outputs.to(device_env.device), prompt_id, tokenizer.trunk.eos_token_id, max_recursion_length,
                                 use_graph=use_graph, return_logits=return_logits)


def generate_ocr(model, tokenizer, encoder_outputs: torch.Tensor, device_env, max_recursion_length: int, prompt_token: str) -> List[str]:
    """ref :143-162: image-encoder outputs -> decoded strings"""
    with torch.inference_mode():
        generated = get_generated_tokens(model, tokenizer, encoder_outputs, device_env, max_recursion_length, prompt_token)
        return [tokenizer.trunk.decode(ids) for ids in generated.tolist()]


# ---------------------------------------------------------------------------------------------- CER / WER
# The reference computes these with jiwer (not installable here): cer / wer = (substitutions + deletions + insertions)
# summed over the batch / reference length summed over the batch, after jiwer transforms. Restated from jiwer's
# documented definitions; reference-side call: utils/ocr_utils.py:32-46 (transforms), :111-140 (cer / wer).
def _edit_distance(a, b) -> int:
    """Levenshtein distance between two token sequences"""
    if len(a) < len(b):
        a, b = b, a
    prev = list(range(len(b) + 1))
    for i, x in enumerate(a, 1):
        cur = [i]
        for j, y in enumerate(b, 1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (x != y)))
        prev = cur
    return prev[-1]


def _remove_pad_words(text: str) -> str:
    """jiwer RemoveSpecificWords("<pad>"): the word is blanked where it stands as a whole word"""
    return re.sub(r'(?<!\S)<pad>(?!\S)', ' ', text)


def _cer_tokens(text: str) -> List[str]:      # RemoveSpecificWords, Strip, ReduceToListOfListOfChars
    return list(_remove_pad_words(text).strip())


def _wer_tokens(text: str) -> List[str]:      # RemoveSpecificWords, RemoveMultipleSpaces, Strip, ReduceToListOfListOfWords
    return [w for w in re.sub(r'\s\s+', ' ', _remove_pad_words(text)).strip().split(' ') if w]


def _error_rate(refs: List[List[str]], hyps: List[List[str]]) -> float:
    total = sum(len(r) for r in refs)
    if total == 0:
        raise ValueError('one or more references are empty strings')     # jiwer raises here; the caller logs and moves on
    return sum(_edit_distance(r, h) for r, h in zip(refs, hyps)) / total


def get_cer_wer_metrics(cer_transforms, wer_transforms, ocr_pretraining_metrics: dict, ocr_predictions, decoded_texts) -> dict:
    """ref :111-140 (the two transform arguments are callables text -> token list; None = the reference's jiwer pipelines)"""
    wer_t = wer_transforms or _wer_tokens
    cer_t = cer_transforms or _cer_tokens
    try:
        ocr_pretraining_metrics['wer'] = _error_rate([wer_t(t) for t in decoded_texts], [wer_t(t) for t in ocr_predictions])
        ocr_pretraining_metrics['cer'] = _error_rate([cer_t(t) for t in decoded_texts], [cer_t(t) for t in ocr_predictions])
    except Exception as e:   # the reference logs and returns what it has
        import logging
        logging.getLogger('ocr').info(f'Encountered exception {e} when computing wer/cer metrics. Length of ground truth texts is '
                                      f'{len(decoded_texts)}, length of generated texts is {len(ocr_predictions)}.')
    return ocr_pretraining_metrics


def get_ocr_metrics(model, tokenizer, image_input, text_input, device_env, max_recursion_length, prompt_token: str
                    ) -> Tuple[Optional[dict], Optional[dict]]:
    """ref :15-108: encode the images, generate greedily (at most as many tokens as the longest target), strip <...>
    tags and newlines from predictions and targets, drop empty pairs, cut predictions to the target's length, CER / WER."""
    with torch.inference_mode():
        m = model.module if hasattr(model, 'module') else model
        image_encoding = m.image_encoder(image_input)
        text_input = text_input.clone()
        text_input[text_input == -100] = tokenizer.trunk.pad_token_id
        sequence_lengths = (text_input != tokenizer.trunk.pad_token_id).sum(dim=1)
        max_recursion_length = min(max_recursion_length, int(sequence_lengths.max().item()))
        ocr_predictions = generate_ocr(m, tokenizer, image_encoding, device_env, max_recursion_length, prompt_token)
        decoded_texts = tokenizer.trunk.batch_decode(text_input)
        clean = lambda t: re.sub(r'<.*?>', '', re.sub('\n', ' ', t))
        ocr_predictions = [clean(t) for t in ocr_predictions]
        decoded_texts = [clean(t) for t in decoded_texts]
        filtered = [(ref, pred) for ref, pred in zip(decoded_texts, ocr_predictions) if ref and pred]
        if not filtered:
            return None, None
        decoded_texts, ocr_predictions = (list(x) for x in zip(*filtered))
        ocr_predictions = [text[0:len(ref)] for text, ref in zip(ocr_predictions, decoded_texts)]
        metrics = get_cer_wer_metrics(None, None, {}, ocr_predictions, decoded_texts)
        sample = {'image': image_input[0], 'original_text': decoded_texts[0], 'reconstructed_text': ocr_predictions[0]}
    return metrics, sample
