#!/usr/bin/env python3
"""EXPERIMENT: decode 8 utterances as ONE chain of batch 8 vs TWO concurrent chains of batch 4 (two decoder engine
handles on two streams) — do the launch-latency phases of two dependent chains overlap on one MI355X?"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import whisper_trtllm_amd as w  # noqa: E402

model = sys.argv[1] if len(sys.argv) > 1 else "whisper-medium.en"
G = int(sys.argv[2]) if len(sys.argv) > 2 else 2
B = int(sys.argv[3]) if len(sys.argv) > 3 else 8
N = 200
cfg = w.synthetic.get_config(model)
weights = w.synthetic.make_weights(cfg, 0)
blob = w.convert.build_decoder_engine(cfg, weights)
hidden = torch.randn(B, cfg["max_source_positions"], cfg["d_model"], device="cuda") * 0.5
one = w.WhisperDecoderEngine(blob, cfg)
many = [w.WhisperDecoderEngine(blob, cfg) for _ in range(G)]
streams = [torch.cuda.Stream() for _ in range(G)]
parts = hidden.chunk(G)


def run_one():
    one.begin(hidden)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    one.steps(N)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / N * 1e3


def run_many():
    for d, s, h in zip(many, streams, parts):
        with torch.cuda.stream(s):
            d.begin(h)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for d, s in zip(many, streams):
        with torch.cuda.stream(s):
            d.steps(N)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / N * 1e3


for _ in range(2):
    a, b = run_one(), run_many()
print(f"{model}: one chain of batch {B}: {a:.3f} ms/step; {G} concurrent chains of batch {B // G}: {b:.3f} ms/step")
with torch.cuda.stream(streams[0]):
    many[0].begin(parts[0])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    many[0].steps(N)
    torch.cuda.synchronize()
    print(f"one chain of batch {B // G} alone: {(time.perf_counter() - t0) / N * 1e3:.3f} ms/step")
