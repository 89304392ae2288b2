"""Cruller fine-tune tasks: RVL-CDIP classification, CORD receipt parsing, DocVQA.
ref: task/task_cruller_finetune_RVLCDIP.py, task_cruller_finetune_CORD.py, task_cruller_finetune_docvqa.py.

The optimisation step is the pretrain step (same HIP kernels, same reducer / optimiser tail); what the reference's
three files add on top of it -- and what this module restates -- is data-side glue:
  * task / prompt tokens and the dataset's special tokens, added to the tokenizer in two stages (the pretrain ones
    before the checkpoint is loaded, the fine-tune ones after) with the embedding table resized each time;
  * `collate_fn(batch)` : PIL images + raw labels -> {"image", "label", "text_target"} with the decoder inputs and the
    labels already shifted against each other;
  * `text_input_to_target` : pads and everything up to and including the prompt-end token are not predicted;
  * `train_step(sample: dict)`.
"""
import logging
from ast import literal_eval
from collections import OrderedDict
from dataclasses import dataclass
from functools import partial
from typing import Any, Dict

import numpy as np
import torch

from ..data import preprocess_ocr_anno, preprocess_text_anno
from ..framework import DeviceEnv, Monitor
from ..utils.json_utils import json2token
from .task_cruller_pretrain import ImagePreprocess, TaskCrullerPretrain, TaskCrullerPretrainCfg

_logger = logging.getLogger(__name__)


@dataclass
class TaskCrullerFinetuneRVLCDIPCfg(TaskCrullerPretrainCfg):
    pass


@dataclass
class TaskCrullerFinetuneCORDCfg(TaskCrullerPretrainCfg):
    pass


@dataclass
class TaskCrullerFinetuneDOCVQACfg(TaskCrullerPretrainCfg):
    pass


class _TaskCrullerFinetune(TaskCrullerPretrain):
    TASK_START_TOKEN = '<s_finetune>'
    PROMPT_END_TOKEN = None          # None: the task start token ends the prompt
    DATASET_TOKENS = ()              # dataset-specific special tokens
    TEXT_ANNO_FN = True
    COLLATE_MAX_LENGTH = 512
    log_phase_name = 'finetune'

    def __init__(self, cfg, device_env: DeviceEnv, monitor: Monitor = None):
        # the parent constructor does what the reference does with `special_tokens_from_pretrain`: tokenizer + model with
        # "<sep/>" and "<s_pretrain>" added, so that a pretrain checkpoint (vocab 50267) loads (ref RVLCDIP :147-162)
        super().__init__(cfg, device_env, monitor)
        self.task_start_token = self.TASK_START_TOKEN
        self.prompt_end_token = self.PROMPT_END_TOKEN or self.task_start_token
        self.text_anno_fn = self.TEXT_ANNO_FN
        self.special_tokens_finetune = ['<sep/>', self.task_start_token, self.prompt_end_token, *self.DATASET_TOKENS]
        preproc_fn = preprocess_text_anno if self.text_anno_fn else preprocess_ocr_anno
        self.anno_preprocess_train = partial(preproc_fn, tokenizer=self.tokenizer.trunk,
                                             max_position_embeddings=self.max_position_embeddings,
                                             task_start_token=self.task_start_token, prompt_end_token=self.prompt_end_token)
        # ToTensor -> Grayscale -> Resize(bicubic, antialias) -> Normalize  (ref docvqa :157-175)
        self.image_preprocess_train = ImagePreprocess(cfg.model.image_encoder.image_size, self.img_mean, self.img_std,
                                                      self.num_image_chs, grayscale=self.num_image_chs == 1)
        # app/train.py assigns the checkpoint to `task.state_dict` and sets `task.resume` (ref app/train.py:156-157);
        # the instance attribute shadows the method exactly as in the reference
        self.resume = False
        self.newly_added_num = 0

    def train_setup(self, num_batches_per_interval: int):
        """ref RVLCDIP :201-236: load the pretrain checkpoint, THEN add the fine-tune tokens and grow the embeddings"""
        ckpt = self.__dict__.get('state_dict')
        if isinstance(ckpt, dict) and len(ckpt):
            _logger.info('Resuming from existing checkpoint.')
            self.model.load_state_dict({k.replace('module.', ''): v for k, v in ckpt.items()})
            del self.__dict__['state_dict']   # give the method back (training_state() / checkpoints need it)
        self.newly_added_num = self.tokenizer.trunk.add_special_tokens(
            {'additional_special_tokens': sorted(set(self.special_tokens_finetune))})
        self.vocab_size = len(self.tokenizer.trunk)
        if self.newly_added_num > 0:
            self.model.text_decoder.trunk.resize_token_embeddings(len(self.tokenizer.trunk))
        super().train_setup(num_batches_per_interval)

    def text_input_to_target(self, text_input: torch.Tensor, ignore_id: int = -100) -> torch.Tensor:
        """ref RVLCDIP :291-300 / docvqa :262-273 (the index of the prompt-end token is taken as the SUM of the matching
        positions, i.e. the sequence is expected to hold it once)"""
        target = text_input.clone()
        target[target == self.tokenizer.trunk.pad_token_id] = ignore_id
        prompt_end_token_id = self.tokenizer.trunk.convert_tokens_to_ids(self.prompt_end_token)
        slice_id = int(torch.nonzero(target == prompt_end_token_id).sum()) + 1
        target[:slice_id] = ignore_id
        return target

    def _tokenize(self, text: str) -> torch.Tensor:
        return self.tokenizer.trunk(text, add_special_tokens=False, return_tensors='pt', max_length=self.COLLATE_MAX_LENGTH,
                                    padding='max_length', truncation=True).input_ids[0]

    def _sequence_for(self, item: Dict[str, Any]) -> str:
        raise NotImplementedError

    def collate_fn(self, batch):
        """list of {"image": PIL / array, <labels>} -> shifted training batch (ref RVLCDIP :302-329)"""
        text_inputs = torch.stack([self._tokenize(self._sequence_for(item)) for item in batch])
        targets = torch.stack([self.text_input_to_target(t) for t in text_inputs])
        images = torch.stack([self.image_preprocess_train(item['image']) for item in batch])
        return {'image': images, 'label': text_inputs[:, :-1], 'text_target': targets[:, 1:]}

    def train_step(self, sample: Dict[str, Any]) -> Dict[str, Any]:
        return self._train_step_shifted(sample['image'], sample['label'], sample['text_target'])


class TaskCrullerFinetuneRVLCDIP(_TaskCrullerFinetune):
    """16-way document classification as generation of one class token: <s_rvlcdip><letter/></s>"""
    TASK_START_TOKEN = '<s_rvlcdip>'
    TEXT_ANNO_FN = False
    COLLATE_MAX_LENGTH = 5
    CLASSES = ('letter', 'form', 'email', 'handwritten', 'advertisement', 'scientific_report', 'scientific_publication',
               'specification', 'file_folder', 'news_article', 'budget', 'invoice', 'presentation', 'questionnaire', 'resume',
               'memo')
    DATASET_TOKENS = ('<s_class>', '</s_class>') + tuple(f'<{c}/>' for c in sorted(CLASSES))

    def __init__(self, cfg, device_env, monitor=None):
        super().__init__(cfg, device_env, monitor)
        self.int2str = dict(enumerate(self.CLASSES))

    def _sequence_for(self, item):
        return self.task_start_token + '<' + self.int2str[int(item['label'])] + '/>' + self.tokenizer.trunk.eos_token


class TaskCrullerFinetuneCORD(_TaskCrullerFinetune):
    """receipt parsing: the ground-truth JSON is flattened by json2token, keys become <s_key>…</s_key> tokens"""
    TASK_START_TOKEN = '<s_cord>'
    # key set of the CORD ground truth (ref CORD :122-177 lists the 54 open / close tokens)
    KEYS = ('cashprice', 'changeprice', 'cnt', 'creditcardprice', 'discount_price', 'discountprice', 'emoneyprice', 'etc',
            'itemsubtotal', 'menu', 'menuqty_cnt', 'menutype_cnt', 'nm', 'num', 'othersvc_price', 'price', 'service_price',
            'sub', 'sub_total', 'subtotal_price', 'tax_price', 'total', 'total_etc', 'total_price', 'unitprice', 'vatyn',
            'void_menu')
    DATASET_TOKENS = tuple(t for k in KEYS for t in (f'<s_{k}>', f'</s_{k}>'))

    def _sequence_for(self, item):
        gt = item['ground_truth']
        gt = literal_eval(gt) if isinstance(gt, str) else gt
        special = list(getattr(self.tokenizer.trunk, 'all_special_tokens', []))
        text, _ = json2token(gt['gt_parse'], special, sort_json_key=False)
        return self.task_start_token + text + self.tokenizer.trunk.eos_token


class TaskCrullerFinetuneDOCVQA(_TaskCrullerFinetune):
    """<s_docvqa><s_question>…</s_question><s_answer>…</s_answer></s>; only the answer is predicted"""
    TASK_START_TOKEN = '<s_docvqa>'
    PROMPT_END_TOKEN = '<s_answer>'
    DATASET_TOKENS = ('<s_question>', '</s_question>', '</s_answer>')

    def _sequence_for(self, item):
        q_and_a = np.random.choice(item['labels'])   # one of the annotated question/answer strings (ref docvqa :287)
        return '<s_docvqa>' + str(q_and_a) + self.tokenizer.trunk.eos_token
