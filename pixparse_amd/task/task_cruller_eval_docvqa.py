"""DocVQA and CORD evaluation tasks (ref: task/task_cruller_eval_docvqa.py, task/task_cruller_eval_cord.py).

Both reference loops decode ONE sample at a time from a multi-token prompt and carry the generation as a STRING: the argmax token
is decoded and appended to `current_string`, which is re-tokenised and pushed through the whole decoder again for the next token
(docvqa :279-297, cord :345-372).  `generate_string` keeps that contract -- the string, the "</s>" stop rule, 512 steps at most, the
re-tokenisation after every token -- on the KV-cache decode path: the prompt goes through one prefill pass
(`Cruller.decode_prefill`), every further token through one `decode_step`; whenever re-tokenising the string does NOT reproduce
"previous ids + the new id" (a tokenizer merging across the boundary) the cache is rebuilt from the re-tokenised ids, so the ids the
decoder sees are at every step exactly the ids the reference would feed."""
import logging
from ast import literal_eval
from dataclasses import dataclass
from typing import List

import numpy as np
import torch

from ..framework import DeviceEnv, Monitor
from ..utils.json_utils import json2token, token2json
from ..utils.metrics import JSONParseEvaluator, average_normalized_levenshtein_similarity
from .task_cruller_eval_ocr import TaskCrullerEvalOCR, TaskCrullerEvalOCRCfg
from .task_cruller_finetune import TaskCrullerFinetuneCORD, TaskCrullerFinetuneDOCVQA

_logger = logging.getLogger(__name__)
MAX_STEPS = 512      # "maximum number of steps" of both reference loops


def generate_string(model, tokenizer, encoder_output: torch.Tensor, prompt: str, device, max_steps: int = MAX_STEPS, stats: dict = None) -> str:
    """encoder_output [S, D] or [1, S, D] of ONE sample -> prompt + generated text (ends with '</s>' unless max_steps ran out)"""
    tok = tokenizer.trunk
    enc = encoder_output if encoder_output.dim() == 3 else encoder_output.unsqueeze(0)
    current = prompt
    ids: List[int] = tok.encode(current, add_special_tokens=False)
    max_pos = model.max_length
    cap = min(max_pos, len(ids) + max_steps + 1)

    if len(ids) >= cap:                               # a prompt that already fills the learned positions: nothing can be generated
        return current

    def restart(seq: List[int]):
        model.decode_begin(enc, cap)
        if len(seq) > 1:
            model.decode_prefill(torch.tensor([seq[:-1]], dtype=torch.int64, device=device))
        if stats is not None:
            stats['prefills'] = stats.get('prefills', 0) + 1

    restart(ids)
    feed = torch.empty(1, 1, dtype=torch.int64, device=device)
    for _ in range(max_steps):
        if len(ids) >= cap:
            break                                     # the learned positions are exhausted (the reference would raise inside HF here)
        feed.fill_(ids[-1])
        next_id = int(torch.argmax(model.decode_step(feed)[0]).item())
        piece = tok.decode([next_id])
        current += piece
        if piece == '</s>':
            break
        new_ids = tok.encode(current, add_special_tokens=False)
        if new_ids == ids + [next_id]:
            ids = new_ids
        else:                                         # the string no longer tokenises to what was generated: follow the reference's ids
            ids = new_ids
            if len(ids) >= cap:
                break
            restart(ids)
    return current


@dataclass
class TaskCrullerEvalDOCVQACfg(TaskCrullerEvalOCRCfg):
    pass


class TaskCrullerEvalDOCVQA(TaskCrullerEvalOCR):
    """ANLS over {'images', 'questions', 'ground_truth_answers', 'question_ids'} batches (ref docvqa :244-313)"""

    def __init__(self, cfg: TaskCrullerEvalDOCVQACfg, device_env: DeviceEnv, monitor: Monitor = None):
        super().__init__(cfg, device_env, monitor)      # tokenizer + model with the two pretrain tokens (ref :97-112)
        self.task_start_token = '<s_docvqa>'
        self.prompt_end_token = '<s_answer>'
        special = ['<sep/>', self.task_start_token, self.prompt_end_token, *TaskCrullerFinetuneDOCVQA.DATASET_TOKENS]
        if self.tokenizer.trunk.add_special_tokens({'additional_special_tokens': sorted(set(special))}) > 0:
            self.model.text_decoder.trunk.resize_token_embeddings(len(self.tokenizer.trunk))
        self.vocab_size = len(self.tokenizer.trunk)
        self.gen_stats = {}

    def setup(self):
        super().setup()
        self.all_ground_truths, self.all_predictions, self.acc_list = [], [], []
        self.evaluator = JSONParseEvaluator()

    def collate_fn(self, batch):
        """ref :244-268"""
        return {'images': torch.stack([self.image_preprocess_eval(item['image']) for item in batch]),
                'questions': [item['labels']['question'] for item in batch],
                'ground_truth_answers': [item['labels']['answers'] for item in batch],
                'image_ids': [item['image_id'] for item in batch],
                'question_ids': [item['question_id'] for item in batch]}

    def step(self, batch):
        metrics = {}
        dev = self.device_env.device
        with torch.inference_mode():
            image_outputs = self.model.image_encoder(batch['images'].to(dev)).clone()     # [B, S, D]: the buffer is reused by later encodes
            for output, question, answers in zip(image_outputs, batch['questions'], batch['ground_truth_answers']):
                self.all_ground_truths.append(answers)
                prompt = self.task_start_token + '<s_question>' + question + '</s_question>' + '<s_answer>'
                text = generate_string(self.model, self.tokenizer, output, prompt, dev, stats=self.gen_stats)
                predicted_json = token2json(text)
                self.all_predictions.append(predicted_json['answer'] if 'answer' in predicted_json else '')
        return metrics

    def average_metrics(self, metrics: dict):
        return {'ANLS': average_normalized_levenshtein_similarity(ground_truth=self.all_ground_truths, predicted_answers=self.all_predictions)}


@dataclass
class TaskCrullerEvalCORDCfg(TaskCrullerEvalOCRCfg):
    pass


class TaskCrullerEvalCORD(TaskCrullerEvalOCR):
    """nTED accuracy + field-level F1 of the generated receipt JSON (ref cord :296-385)"""
    COLLATE_MAX_LENGTH = 512

    def __init__(self, cfg: TaskCrullerEvalCORDCfg, device_env: DeviceEnv, monitor: Monitor = None):
        super().__init__(cfg, device_env, monitor)
        self.task_start_token = '<s_cord>'
        self.prompt_end_token = self.task_start_token
        special = ['<sep/>', self.task_start_token, *TaskCrullerFinetuneCORD.DATASET_TOKENS]
        if self.tokenizer.trunk.add_special_tokens({'additional_special_tokens': sorted(set(special))}) > 0:
            self.model.text_decoder.trunk.resize_token_embeddings(len(self.tokenizer.trunk))
        self.vocab_size = len(self.tokenizer.trunk)
        self.gen_stats = {}

    def setup(self):
        super().setup()
        self.all_ground_truths, self.all_predictions, self.acc_list = [], [], []
        self.evaluator = JSONParseEvaluator()

    def text_input_to_target(self, text_input, ignore_id=-100):
        """ref :283-294"""
        target = text_input.clone()
        target[target == self.tokenizer.trunk.pad_token_id] = ignore_id
        prompt_end_token_id = self.tokenizer.trunk.convert_tokens_to_ids(self.prompt_end_token)
        slice_id = int(torch.nonzero(target == prompt_end_token_id).sum()) + 1
        target[:slice_id] = ignore_id
        return target

    def collate_fn(self, batch):
        """ref :296-333: {'image', 'ground_truth' (str of a dict with 'gt_parse')} items -> images + shifted token sequences"""
        tok = self.tokenizer.trunk
        seqs = []
        for item in batch:
            gt = item['ground_truth']
            gt = literal_eval(gt) if isinstance(gt, str) else gt
            text, _ = json2token(gt['gt_parse'], list(getattr(tok, 'all_special_tokens', [])), sort_json_key=False)
            seqs.append(tok(self.task_start_token + text + tok.eos_token, add_special_tokens=False, return_tensors='pt',
                            max_length=self.COLLATE_MAX_LENGTH, padding='max_length', truncation=True).input_ids[0])
        text_inputs = torch.stack(seqs)
        targets = torch.stack([self.text_input_to_target(t) for t in text_inputs])
        images = torch.stack([self.image_preprocess_eval(item['image']) for item in batch])
        return {'image': images, 'label': text_inputs[:, :-1], 'text_target': targets[:, 1:]}

    def step(self, batch):
        metrics = {}
        dev = self.device_env.device
        acc = None
        for image, label in zip(batch['image'], batch['label']):
            ground_truth = token2json(self.tokenizer.trunk.decode(label))
            with torch.inference_mode():
                output = self.model.image_encoder(image.unsqueeze(0).to(dev))
                text = generate_string(self.model, self.tokenizer, output, '<s_cord>', dev, stats=self.gen_stats)
                predicted_json = token2json(text)
            self.all_predictions.append(predicted_json)
            self.all_ground_truths.append(ground_truth)
            acc = self.evaluator.cal_acc(predicted_json, ground_truth)
            self.acc_list.append(acc)
        metrics['batch_accuracy'] = acc
        return metrics

    def average_metrics(self, metrics: dict):
        avg_accuracy = float(np.mean(self.acc_list))
        f1 = self.evaluator.cal_f1(self.all_predictions, self.all_ground_truths)
        self.all_ground_truths, self.all_predictions, self.acc_list = [], [], []
        return {'average_accuracy': avg_accuracy, 'f1_score': f1}
