"""Greedy generation for the eval tasks / train-time OCR metric.
ref: utils/ocr_utils.py:143-197 (`generate_ocr`, `get_generated_tokens`) and :200-222 (`get_next_token`).

Same signatures and the same loop semantics as the reference -- every sequence starts from the prompt token, a sample
that has produced eos keeps being extended until ALL samples have, the token that completes the last sample is not
appended -- but where the reference re-runs the whole decoder on the growing prefix for every token, this goes through
`Cruller.decode_begin` / `decode_step`: cross-attention K/V projected once, self-attention K/V cached, one token of new
work per step (skinny HBM-bound projections + split-KV single-query attention), and the step -- free of host-visible
state -- is captured once in a hipGraph and replayed (`Cruller.generate_greedy`)."""
from typing import List, Tuple

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
