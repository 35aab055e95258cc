"""Greedy-search session pieces with the reference's surface (examples/whisper/run.py:150-227).

`run.py` imports the logits processors / stopping criteria from the bundled transformers
(generation/logits_process.py:1281-1328, stopping_criteria.py:44-72).  They are re-stated here on torch
tensors so the Session path has no transformers dependency; the fast path implements the same rules on the
device (csrc/kernels_decoder.hip greedy_select_kernel).
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch


class LogitsProcessorList(list):
    def __call__(self, input_ids, scores):
        for proc in self:
            scores = proc(input_ids, scores)
        return scores


class SuppressTokensLogitsProcessor:
    def __init__(self, suppress_tokens):
        self.suppress_tokens = list(suppress_tokens)

    def __call__(self, input_ids, scores):
        scores[:, self.suppress_tokens] = -float("inf")
        return scores


class SuppressTokensAtBeginLogitsProcessor:
    def __init__(self, begin_suppress_tokens, begin_index):
        self.begin_suppress_tokens, self.begin_index = list(begin_suppress_tokens), begin_index

    def __call__(self, input_ids, scores):
        if input_ids.shape[1] == self.begin_index:
            scores[:, self.begin_suppress_tokens] = -float("inf")
        return scores


class ForceTokensLogitsProcessor:
    def __init__(self, force_token_map):
        self.force_token_map: Dict[int, int] = dict(force_token_map)

    def __call__(self, input_ids, scores):
        tok = self.force_token_map.get(input_ids.shape[-1], None)
        if tok is not None:
            scores[:, :] = -float("inf")
            scores[:, tok] = 0
        return scores


class MaxLengthCriteria:
    def __init__(self, max_length: int, max_position_embeddings: Optional[int] = None):
        self.max_length = max_length

    def __call__(self, input_ids, scores=None) -> bool:
        return input_ids.shape[-1] >= self.max_length


class StoppingCriteriaList(list):
    def __call__(self, input_ids, scores=None) -> bool:
        return any(c(input_ids, scores) for c in self)


def get_logits_processor(config: dict, input_ids_seq_length: int) -> LogitsProcessorList:
    """run.py:150-162."""
    procs = LogitsProcessorList()
    procs.append(SuppressTokensLogitsProcessor(config["suppress_tokens"]))
    begin_index = input_ids_seq_length if config["forced_bos_token_id"] is None else input_ids_seq_length + 1
    begin_index += config["forced_decoder_ids"][-1][0]
    procs.append(SuppressTokensAtBeginLogitsProcessor(config["begin_suppress_tokens"], begin_index))
    procs.append(ForceTokensLogitsProcessor(config["forced_decoder_ids"]))
    return procs


def get_stopping_criteria(config: dict) -> StoppingCriteriaList:
    """run.py:164-169."""
    return StoppingCriteriaList([MaxLengthCriteria(max_length=config["max_length"])])


def greedy_search(model, encoder_outputs, input_ids, logits_processor=None, stopping_criteria=None,
                  pad_token_id=None, eos_token_id=None):
    """Token-at-a-time greedy loop over a `model(ids, enc, past) -> (logits, past)` callable — run.py:171-227."""
    eos = torch.tensor([eos_token_id], device=input_ids.device)
    unfinished = torch.ones(input_ids.shape[0], dtype=torch.int32, device=input_ids.device)
    past = None
    while True:
        output, past = model(input_ids[:, -1:], encoder_outputs, past)
        scores = logits_processor(input_ids, output[:, -1, :])
        nxt = torch.argmax(scores, dim=-1).to(input_ids.dtype)
        nxt = nxt * unfinished + pad_token_id * (1 - unfinished)
        input_ids = torch.cat([input_ids, nxt[:, None]], dim=-1)
        unfinished = unfinished * (nxt != eos).to(torch.int32)
        if unfinished.max() == 0 or stopping_criteria(input_ids, None):
            break
    return input_ids
