#!/usr/bin/env python3
"""Soak check (GPU box): six engine open/close cycles, repeated decodes (bitwise equal every time), batch-size changes that
re-plan the workspace and the step graph, and a device-memory leak check across the cycles."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import whisper_trtllm_amd as w
cfg = w.synthetic.get_config("whisper-small.en")
weights = w.synthetic.make_weights(cfg, 0)
eb, db = w.convert.build_encoder_engine(cfg, weights), w.convert.build_decoder_engine(cfg, weights)
mel = torch.from_numpy(w.synthetic.make_mel(cfg, 0, 8)).cuda()
ref = None
free0 = None
for rep in range(6):
    enc, dec = w.WhisperEncoderEngine(eb), w.WhisperDecoderEngine(db, cfg)
    for it in range(3):
        ids = dec.generate(enc(mel), max_length=64)
        if ref is None:
            ref = ids.clone()
        assert torch.equal(ids, ref), (rep, it)
    for B in (1, 3, 16, 5):   # batch changes re-plan the workspace and the step graph
        out = dec.generate(enc(mel.repeat(2, 1, 1)[:B]), max_length=32)
        assert torch.equal(out[: min(B, 8)], ref[: min(B, 8), :32]), (rep, B)
    del enc, dec
    torch.cuda.synchronize()
    free, total = torch.cuda.mem_get_info()
    if free0 is None:
        free0 = free
    print(f"rep {rep}: free {free / 2**30:.2f} GiB (first {free0 / 2**30:.2f})", flush=True)
assert abs(free - free0) < 64 * 2**20, "device memory is leaking across engine open/close"
print("soak ok")
