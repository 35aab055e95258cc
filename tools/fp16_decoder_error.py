#!/usr/bin/env python3
"""GPU box: logits error of the fp16 decoder engine against the oracle's fp16_engine mode (and against the fp32 oracle) per golden case --
the numbers LOGITS_TOL in tests/test_gpu_fp16_decoder.py is set from.  Imports the oracle: a measurement helper, not product code."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import torch
import whisper_trtllm_amd as wt
import cpu_ref
from conftest import load_case
for case in sys.argv[1:] or ["toy-short_b3", "toy-short-eos1_b3", "toy_b1", "tiny_b2", "toy-wide_b2"]:
    z, cfg, weights, mel = load_case(case)
    enc = wt.WhisperEncoderEngine(wt.convert.build_encoder_engine(cfg, weights))
    dec = wt.WhisperDecoderEngine(wt.convert.build_decoder_engine(cfg, weights, precision="float16"), cfg)
    hidden = enc(torch.from_numpy(mel).cuda())
    B, V, ml = mel.shape[0], cfg["vocab_size"], cfg["max_length"]
    trace = torch.zeros(B, ml - 1, V, dtype=torch.float32, device="cuda")
    ids = dec.generate(hidden, logits_trace=trace).cpu()
    W16 = cpu_ref.to_torch(cpu_ref.fp16_engine_weights(weights, decoder=True))
    with torch.no_grad():
        ids16, lg16 = cpu_ref.greedy_search(W16, cfg, hidden.cpu(), return_logits=True, fp16_engine=True)
        ids32, lg32 = cpu_ref.greedy_search(cpu_ref.to_torch(weights), cfg, hidden.cpu(), return_logits=True)
    n = lg16.shape[1]
    e16 = float((trace[:, :n].cpu() - lg16).abs().max())
    n32 = min(n, lg32.shape[1])
    e32 = float((trace[:, :n32].cpu() - lg32[:, :n32]).abs().max())
    print(f"{case}: |engine - fp16 oracle| {e16:.3e}   |engine - fp32 oracle| {e32:.3e}   logits scale {float(lg16.abs().max()):.2f}   ids equal fp16 oracle {bool(ids.shape == ids16.shape and torch.equal(ids, ids16.to(ids.dtype)))}", flush=True)
