#!/usr/bin/env python3
"""Decode phase alone (all max_length-1 steps of one batch, graph-replayed) -- ms per step, for same-box A/B runs of the tuning
switches (WT_TUNING=1 WT_PREFETCH=0|1 WT_PREFETCH_BLOCKS=.. etc., one process per setting).  Also prints the graph-replayed time of
every launch kind of a decoder layer (wt_decoder_time_kernel)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import whisper_trtllm_amd as w  # noqa: E402

model = sys.argv[1] if len(sys.argv) > 1 else "whisper-medium.en"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
kinds = len(sys.argv) > 4 and sys.argv[4] == "kinds"
cfg = w.synthetic.get_config(model)
weights = w.synthetic.make_weights(cfg, 0)
enc = w.WhisperEncoderEngine(w.convert.build_encoder_engine(cfg, weights))
dec = w.WhisperDecoderEngine(w.convert.build_decoder_engine(cfg, weights), cfg)
mel = torch.from_numpy(w.synthetic.make_mel(cfg, 0, B)).cuda()
hidden = enc(mel)
n = cfg["max_length"] - 1
ids0 = None
for r in range(reps + 1):
    dec.begin(hidden)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    dec.steps(n)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ids = dec.read_ids(cfg["max_length"]).cpu()
    if ids0 is None:
        ids0 = ids
    assert torch.equal(ids, ids0)
    if r:
        print(f"{model} B={B}: {n} steps {dt * 1e3:8.2f} ms  {dt / n * 1e3:.4f} ms/step", flush=True)
for r in range(2):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    dec.generate(enc(mel))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"enc + generate: {dt * 1e3:8.2f} ms  ({30 * B / dt:.1f} audio-s/s)", flush=True)
print("ids checksum", int(ids0.long().sum()))
if kinds:
    dec.begin(hidden)
    dec.steps(n)
    dec.poll()
    for k in ("qkv", "self_attn", "pair", "cross_attn", "cross_out", "fc1", "fc2"):
        print(f"  {k:10s} {dec.time_kernel(k, 20):6.2f} us")
