#!/usr/bin/env python3
"""WER on a `librispeech.cache` of (log-mel [80,3000], text) pairs — the reference's examples/whisper/cal_wer.py flow
(:249-287) on the batched fast path.  The cache comes from examples/whisper/get_LibriSpeech.py (this repo's, or the reference's:
same format).  No LibriSpeech or `whisper-*.en` checkpoint exists on the build/GPU boxes, so the script is GPU-tested end to end
on artefacts the tests write themselves (tests/test_gpu_session.py::test_cal_wer_script_end_to_end, also under torchrun with
two ranks); the README WER table stays to be reproduced on a box that has the data.

Multi-GPU (BASELINE config 5, utterances sharded over the GPUs of one node -- no collective on the data path):
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 examples/whisper/cal_wer.py ...
every rank decodes a contiguous shard of the cache on its own GPU, hypotheses are gathered on the host, rank 0 prints the WER.

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
    parser.add_argument("--batching", choices=["continuous", "sorted", "dataset"], default="continuous",
                        help="continuous (default): dataset order through the continuous mode -- `--batch` decode slots stay busy, every "
                             "utterance stops at its own EOS (as the reference's one-clip-at-a-time loop, cal_wer.py:249-287) and its slot is "
                             "refilled on the device; sorted: length-aware batches (utterances ordered by the audio duration recovered from the "
                             "log-mel's trailing padding); dataset: contiguous batches, each decoding to its longest row")
    parser.add_argument("--workers", type=int, default=2, help="engine pairs per GPU, each on its own stream and host thread (1 = one batch in "
                        "flight at a time; 2 fills the decode's launch-latency gaps with the other batch's work: ~1.4x)")
    parser.add_argument("--dist-backend", type=str, default=None, help="nccl (= RCCL) | gloo; default: nccl when every rank has its own GPU")
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
    rank, world, device, dist = tensorrt_llm.sharding.init_from_env(args.dist_backend)
    torch.cuda.set_device(device)
    with open(os.path.join(args.engine_dir, "config.pkl"), "rb") as f:
        config = pickle.load(f)
    pipe = tensorrt_llm.WhisperPipeline(open(os.path.join(args.engine_dir, "WhisperEncoder.engine"), "rb").read(),
                                        open(os.path.join(args.engine_dir, "WhisperDecoder.engine"), "rb").read(), config, workers=max(1, args.workers))
    tok = WhisperTokenDecoder.from_dir(args.whisper)
    eos = config["eos_token_id"]
    with open(args.cache, "rb") as f:
        dataset = pickle.load(f)
    indexed = []
    if args.batching == "continuous":   # this rank's contiguous shard in arrival order; nothing to sort, no row waits for a neighbour
        begin, end = tensorrt_llm.sharding.utterance_shard(len(dataset), world, rank)
        for c0 in range(begin, end, 512):      # host -> device in chunks of 512 utterances (a log-mel is 0.96 MB)
            idx = list(range(c0, min(c0 + 512, end)))
            mels = torch.stack([torch.as_tensor(dataset[i][0], dtype=torch.float32) for i in idx]).cuda()
            rows = pipe.transcribe_continuous(mels, slots=args.batch, chunk=args.batch)
            indexed += list(zip(idx, tok.batch_decode([r.tolist() for r in rows], skip_special_tokens=True)))
        groups = []
    elif args.batching == "sorted":   # every rank gets the same mix of long and short batches (round-robin over the sorted batches)
        lengths = [tensorrt_llm.audio.valid_frames(torch.as_tensor(m, dtype=torch.float32))[0] for m, _ in dataset]
        groups = tensorrt_llm.sharding.length_sorted_batches(lengths, args.batch, world, rank)
    else:
        begin, end = tensorrt_llm.sharding.utterance_shard(len(dataset), world, rank)   # this rank's contiguous shard
        groups = [list(range(b0, b1)) for b0, b1 in tensorrt_llm.sharding.batches(begin, end, args.batch)]
    for c0 in range(0, len(groups), 64):      # host -> device in chunks of 64 batches (a log-mel is 0.96 MB), decoded `workers` at a time
        chunk = groups[c0:c0 + 64]
        mels = [torch.stack([torch.as_tensor(dataset[i][0], dtype=torch.float32) for i in g]).cuda() for g in chunk]
        for g, ids in zip(chunk, pipe.transcribe(mels)):
            # a row of a batch pads on behind its own EOS until the longest row stops; the reference decodes one clip at a time and
            # ends AT the EOS (run.py:219-226) -- cut there (only matters when pad_token_id is not a special token)
            rows = [r[:r.index(eos, 1) + 1] if eos in r[1:] else r for r in ids.cpu().tolist()]
            indexed += list(zip(g, tok.batch_decode(rows, skip_special_tokens=True)))
    indexed = sorted(tensorrt_llm.sharding.gather_objects(indexed, dist))            # back to dataset order on the host
    hypotheses = [h for _, h in indexed]
    references = [t for _, t in dataset]
    if rank == 0:
        assert len(hypotheses) == len(dataset)
        normalizer = get_normalizer(args.whisper)
        wer = word_error_rate([normalizer(t) for t in references], [normalizer(t) for t in hypotheses])
        print(f"WER: {wer * 100:.2f} %  ({len(dataset)} utterances, {world} rank(s))")
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
