#!/usr/bin/env python3
"""Run the engines built by build_encoder.py / build_decoder.py — the reference's examples/whisper/run.py flow.

Two decode paths over the same C-ABI:
  * Session path (default with --session): `WhisperEncoder` / `WhisperDecoder` wrappers + `greedy_search`, i.e. the
    reference's per-token `Session.run` protocol with by-value caches (run.py:57-227), batch 1;
  * fast path (default): resident in-place KV cache, on-device greedy loop, utterance batches of 8
    (`WhisperEncoderEngine` / `WhisperDecoderEngine`).
Inputs: `--audio a.wav ...` (16 kHz mono; log-mel on the GPU front-end, transcripts printed when `--whisper` is a checkpoint
directory with vocab.json) or, without audio on the box, `--synthetic N` seeded synthetic 80x3000 log-mels (outputs are token ids).
Multi-GPU: under `python -m torch.distributed.run --nproc-per-node N ...` the fast path shards the utterances over the ranks
(one process per GPU, full weight replica each, no data-path collective); ids are gathered on the host and rank 0 reports."""
import argparse
import os
import pickle
import time

import torch

from _common import ROOT  # noqa: F401

import whisper_trtllm_amd as tensorrt_llm
from whisper_trtllm_amd import trt
from whisper_trtllm_amd.generation import get_logits_processor, get_stopping_criteria, greedy_search
from whisper_trtllm_amd.runtime import Session, TensorInfo, _scoped_stream

_trt_to_torch_dtype_dict = {trt.float16: torch.float16, trt.float32: torch.float32, trt.int32: torch.int32, trt.int8: torch.int8}
_torch_to_trt_dtype_dict = {v: k for k, v in _trt_to_torch_dtype_dict.items()}


def parse_arguments():
    parser = argparse.ArgumentParser()
    parser.add_argument("--whisper", type=str, default="synthetic:whisper-tiny.en")
    parser.add_argument("--engine_precision", type=str, default="float32")
    parser.add_argument("--log_level", type=str, default="error")
    parser.add_argument("--engine_dir", type=str, default="whisper_outputs")
    parser.add_argument("--compare", action="store_true", help="also decode through the Session path and compare ids")
    parser.add_argument("--session", action="store_true", help="decode through the per-token Session protocol only")
    parser.add_argument("--synthetic", type=int, default=8, help="number of synthetic utterances")
    parser.add_argument("--synthetic_start", type=int, default=0, help="index of the first synthetic utterance (synthetic.make_mel)")
    parser.add_argument("--audio", nargs="+", default=None, help="16 kHz mono audio files to transcribe instead (log-mel by the GPU "
                        "front-end, i.e. what hf_processor does in the reference, run.py:267); with --whisper <checkpoint dir> the ids "
                        "are also decoded to text (vocab.json)")
    parser.add_argument("--max_length", type=int, default=None)
    parser.add_argument("--batching", choices=["continuous", "sorted", "dataset"], default="sorted",
                        help="fast path: continuous (arrival order, 8 decode slots refilled on the device the step an utterance stops -- every "
                             "utterance ends at its own EOS like the reference's batch-1 loop, run.py:219-226), length-aware batches (by the audio "
                             "duration recovered from the log-mel's trailing padding) or dataset-order batches")
    parser.add_argument("--workers", type=int, default=2, help="fast path: engine pairs per GPU, each on its own stream and host thread")
    parser.add_argument("--dist-backend", type=str, default=None, help="nccl (= RCCL) | gloo; default: nccl when every rank has its own GPU")
    parser.add_argument("--dump_ids", type=str, default=None, help="rank 0 writes the fast-path token ids as JSON (tests)")
    return parser.parse_args()


def trim_after_eos(row, eos, pad):
    """A batched fast-path row is padded with `pad` up to the longest row of its batch; a batch-1 Session row stops at its own EOS
    (run.py:219-226).  Cut a row behind its first EOS so the two are comparable (pad == eos for the .en checkpoints)."""
    row = list(row)
    if eos in row[1:]:
        row = row[:row.index(eos, 1) + 1]
    while len(row) > 1 and row[-1] == pad and pad != eos:
        row.pop()
    return row


class WhisperEncoder:
    """Engine wrapper with the reference's call signature (run.py:57-93): mel [1,80,3000] -> hidden [1,1500,d]."""

    def __init__(self, args=None, config=None):
        with open(os.path.join(args.engine_dir, "WhisperEncoder.engine"), "rb") as f:
            self.session = Session.from_serialized_engine(f.read())
        frames = 2 * config["max_source_positions"]
        outputs_shape = self.session.infer_shapes([TensorInfo("data", trt.float32, (1, config["num_mel_bins"], frames)),
                                                   TensorInfo("length", trt.float32, (1,))])
        self.inputs = {"data": torch.rand(1, config["num_mel_bins"], frames).cuda(), "length": torch.Tensor([1.0]).cuda()}
        self.outputs = {o.name: torch.zeros(*o.shape, dtype=_trt_to_torch_dtype_dict[o.dtype]).cuda() for o in outputs_shape}

    def __call__(self, input):
        self.inputs["data"] = input
        with _scoped_stream() as stream:
            ok = self.session.run(self.inputs, self.outputs, stream)
        assert ok
        return self.outputs["hidden_states"].clone()


class WhisperDecoder:
    """Per-token decoder wrapper with the by-value cache protocol of run.py:95-148 (masks carry lengths only)."""

    def __init__(self, args=None, config=None):
        self.config = config
        with open(os.path.join(args.engine_dir, "WhisperDecoder.engine"), "rb") as f:
            self.session = Session.from_serialized_engine(f.read())

    def __call__(self, decoder_input_ids, encoder_outputs, past_key_values):
        config = self.config
        S = config["max_source_positions"]
        inputs = {"data": decoder_input_ids.to(dtype=torch.int32, device="cuda"),
                  "length": torch.Tensor([1.0]).to(dtype=torch.int32).cuda(),
                  "encoder_hidden_states": encoder_outputs.to(dtype=torch.float32, device="cuda")}
        if past_key_values is None:
            L, H = config["decoder_layers"], config["decoder_attention_heads"]
            dh = config["d_model"] // H
            inputs["self_past_key"] = torch.rand(L, H, 1, dh).cuda()
            inputs["self_past_value"] = torch.rand(L, H, 1, dh).cuda()
            inputs["cross_past_key"] = torch.rand(L, H, S, dh).cuda()
            inputs["cross_past_value"] = torch.rand(L, H, S, dh).cuda()
            inputs["past_self_cache_mask"] = torch.rand(1).cuda()
            inputs["past_cross_cache_mask"] = torch.rand(1).cuda()
        else:
            for name, t in zip(("self_past_key", "self_past_value", "cross_past_key", "cross_past_value"), past_key_values):
                inputs[name] = t.to(dtype=torch.float32, device="cuda")
            inputs["past_self_cache_mask"] = torch.rand(int(1 + past_key_values[0].shape[2]), dtype=torch.float32).cuda()
            inputs["past_cross_cache_mask"] = torch.rand(int(1 + S), dtype=torch.float32).cuda()
        outputs_shape = self.session.infer_shapes([TensorInfo(k, _torch_to_trt_dtype_dict[v.dtype], tuple(v.shape)) for k, v in inputs.items()])
        outputs = {o.name: torch.zeros(*o.shape, dtype=_trt_to_torch_dtype_dict[o.dtype]).cuda() for o in outputs_shape}
        with _scoped_stream() as stream:
            ok = self.session.run(inputs, outputs, stream)
        assert ok
        return outputs["hidden_states"], (outputs["next_self_keys"], outputs["next_self_values"],
                                          outputs["next_cross_keys"], outputs["next_cross_values"])


def decode_with_sessions(whisperencoder, whisperdecoder, config, mel):
    """One utterance through the reference's loop (run.py:266-284)."""
    encoder_outputs = whisperencoder(mel)
    input_ids = torch.Tensor([[config["decoder_start_token_id"]]]).to(dtype=torch.int32).cuda()
    return greedy_search(model=whisperdecoder, encoder_outputs=encoder_outputs, input_ids=input_ids,
                         logits_processor=get_logits_processor(config, input_ids.shape[-1]),
                         stopping_criteria=get_stopping_criteria(config),
                         pad_token_id=config["pad_token_id"], eos_token_id=config["eos_token_id"])


if __name__ == "__main__":
    args = parse_arguments()
    tensorrt_llm.logger.set_level(args.log_level)
    rank, world, device, dist = tensorrt_llm.sharding.init_from_env(args.dist_backend)
    torch.cuda.set_device(device)
    if world > 1 and (args.session or args.compare):
        raise SystemExit("the Session path is the reference's batch-1 single-GPU protocol: run --session / --compare without torchrun")
    with open(os.path.join(args.engine_dir, "config.pkl"), "rb") as f:
        config = pickle.load(f)
    if args.max_length:
        config["max_length"] = args.max_length
    name = config.get("name", "whisper-tiny.en")
    if args.audio:
        import numpy as np
        from get_LibriSpeech import N_SAMPLES, read_audio
        frontend = tensorrt_llm.audio.LogMelFrontend()
        wav = np.zeros((len(args.audio), N_SAMPLES), dtype=np.float32)
        for j, path in enumerate(args.audio):
            a = read_audio(path)[:N_SAMPLES]
            wav[j, :len(a)] = a
        mels = list(frontend(torch.from_numpy(wav).cuda()).split(1))
    else:
        mels = [torch.from_numpy(tensorrt_llm.synthetic.make_mel(config, index=args.synthetic_start + i, batch=1)).cuda() for i in range(args.synthetic)]
    results = {}
    if not args.session:
        pipe = tensorrt_llm.WhisperPipeline(open(os.path.join(args.engine_dir, "WhisperEncoder.engine"), "rb").read(),
                                            open(os.path.join(args.engine_dir, "WhisperDecoder.engine"), "rb").read(), config, workers=max(1, args.workers))
        begin, end = tensorrt_llm.sharding.utterance_shard(len(mels), world, rank)       # this rank's contiguous shard
        if args.batching == "continuous":
            groups = None
        elif args.batching == "sorted":
            lengths = [tensorrt_llm.audio.valid_frames(m)[0] for m in mels]
            groups = tensorrt_llm.sharding.length_sorted_batches(lengths, 8, world, rank)
        else:
            begin, end = tensorrt_llm.sharding.utterance_shard(len(mels), world, rank)   # this rank's contiguous shard
            groups = [list(range(b0, b1)) for b0, b1 in tensorrt_llm.sharding.batches(begin, end, 8)]
        for _ in range(2):  # the first pass is the warm-up, as in run.py:260
            if dist is not None:
                dist.barrier()
            torch.cuda.synchronize()
            t0 = time.time()
            indexed = []
            if groups is None:
                rows = pipe.transcribe_continuous(torch.cat(mels[begin:end]), slots=8, chunk=8) if end > begin else []
                indexed = list(zip(range(begin, end), [r.tolist() for r in rows]))
            else:
                for g, ids in zip(groups, pipe.transcribe([torch.cat([mels[i] for i in g]) for g in groups])):
                    indexed += list(zip(g, ids.cpu().tolist()))
            torch.cuda.synchronize()
            elapsed = tensorrt_llm.sharding.max_over_ranks(time.time() - t0, dist)
            ids = [row for _, row in sorted(tensorrt_llm.sharding.gather_objects(indexed, dist))]   # dataset order, on the host
            results["fast"] = (elapsed, ids)
        if rank == 0:
            print(f"fast path   : {results['fast'][0]:.3f} s for {len(mels)} x 30 s on {world} rank(s)  ({30 * len(mels) / results['fast'][0]:.1f} audio-s/s)")
            if args.dump_ids:
                import json
                json.dump(results["fast"][1], open(args.dump_ids, "w"))
    if args.session or args.compare:
        whisperencoder, whisperdecoder = WhisperEncoder(args, config), WhisperDecoder(args, config)
        for _ in range(2):
            torch.cuda.synchronize()
            t0 = time.time()
            ids = [decode_with_sessions(whisperencoder, whisperdecoder, config, m)[0].cpu().tolist() for m in mels]
            torch.cuda.synchronize()
            results["session"] = (time.time() - t0, ids)
        print(f"Session path: {results['session'][0]:.3f} s for {len(mels)} x 30 s  ({30 * len(mels) / results['session'][0]:.1f} audio-s/s)")
    if args.compare:
        eos, pad = config["eos_token_id"], config["pad_token_id"]
        a = [trim_after_eos(r, eos, pad) for r in results["fast"][1]]
        b = [trim_after_eos(r, eos, pad) for r in results["session"][1]]
        diff = [(x, y) for x, y in zip(a, b) if x != y]
        print(f"Compare Result: same [{len(a) - len(diff)}], diff [{len(diff)}]")
    if rank == 0:
        for key in results:
            print(key, "ids[0][:16] =", results[key][1][0][:16])
    if rank == 0 and args.audio and os.path.isdir(args.whisper):   # batch_decode(predicted_ids, skip_special_tokens=True) of run.py:287
        tok = tensorrt_llm.text.WhisperTokenDecoder.from_dir(args.whisper)
        for path, ids in zip(args.audio, next(iter(results.values()))[1]):
            print(f"{os.path.basename(path)}: {tok.decode(ids, skip_special_tokens=True)!r}")
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
