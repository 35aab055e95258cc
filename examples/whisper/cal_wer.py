#!/usr/bin/env python3
"""WER on a `librispeech.cache` of (log-mel [80,3000], text) pairs — the reference's examples/whisper/cal_wer.py flow
(:249-287) on the batched fast path.  Needs real `whisper-*.en` engines and the cache produced by the reference's
get_LibriSpeech.py; neither exists on the build/GPU boxes, so this script is exercised only by its unit-tested parts
(tokenizer decode, English normaliser and WER: tests/test_text.py).

The cache is a pickle (as in the reference): only load files you created yourself."""
import argparse
import os
import pickle

import torch

from _common import ROOT  # noqa: F401

import whisper_trtllm_amd as tensorrt_llm
from whisper_trtllm_amd.english import EnglishTextNormalizer
from whisper_trtllm_amd.text import WhisperTokenDecoder, word_error_rate


def parse_arguments():
    parser = argparse.ArgumentParser()
    parser.add_argument("--whisper", type=str, required=True, help="local HF checkpoint dir (vocab.json, normalizer.json)")
    parser.add_argument("--engine_dir", type=str, default="whisper_outputs")
    parser.add_argument("--cache", type=str, default="librispeech.cache")
    parser.add_argument("--batch", type=int, default=8)
    parser.add_argument("--log_level", type=str, default="error")
    return parser.parse_args()


def get_normalizer(whisper_dir):
    """The reference normalises with whisper's EnglishTextNormalizer (cal_wer.py:11, 281); `whisper_trtllm_amd.english` restates
    it.  The British->American spelling table ships with the checkpoint (normalizer.json); without it spellings are left alone."""
    import json
    path = os.path.join(whisper_dir, "normalizer.json")
    return EnglishTextNormalizer(json.load(open(path, encoding="utf-8")) if os.path.exists(path) else None)


if __name__ == "__main__":
    args = parse_arguments()
    tensorrt_llm.logger.set_level(args.log_level)
    torch.cuda.set_device(0)
    with open(os.path.join(args.engine_dir, "config.pkl"), "rb") as f:
        config = pickle.load(f)
    enc = tensorrt_llm.WhisperEncoderEngine(open(os.path.join(args.engine_dir, "WhisperEncoder.engine"), "rb").read())
    dec = tensorrt_llm.WhisperDecoderEngine(open(os.path.join(args.engine_dir, "WhisperDecoder.engine"), "rb").read(), config)
    tok = WhisperTokenDecoder.from_dir(args.whisper)
    with open(args.cache, "rb") as f:
        dataset = pickle.load(f)
    hypotheses, references = [], []
    for i in range(0, len(dataset), args.batch):
        chunk = dataset[i:i + args.batch]
        mel = torch.stack([torch.as_tensor(m, dtype=torch.float32) for m, _ in chunk]).cuda()
        ids = dec.generate(enc(mel)).cpu().tolist()
        hypotheses += tok.batch_decode(ids, skip_special_tokens=True)
        references += [t for _, t in chunk]
    normalizer = get_normalizer(args.whisper)
    wer = word_error_rate([normalizer(t) for t in references], [normalizer(t) for t in hypotheses])
    print(f"WER: {wer * 100:.2f} %")
