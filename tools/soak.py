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

# ---- multi-worker soak: 4 workers (clones on one copy of the weights), 240 small batches of changing size, three rounds; every result
# bitwise that of one engine pair; device memory back to where it was
cfg2 = w.synthetic.get_config("whisper-tiny.en")
cfg2["max_length"] = 24
weights2 = w.synthetic.make_weights(cfg2, 1)
eb2, db2 = w.convert.build_encoder_engine(cfg2, weights2), w.convert.build_decoder_engine(cfg2, weights2)
enc1, dec1 = w.WhisperEncoderEngine(eb2), w.WhisperDecoderEngine(db2, cfg2)
sizes = [1 + (7 * i) % 16 for i in range(240)]
mels = [torch.from_numpy(w.synthetic.make_mel(cfg2, index=3 * i, batch=b)).cuda() for i, b in enumerate(sizes[:24])]
batches = [mels[i % 24][: sizes[i]] if sizes[i] <= mels[i % 24].shape[0] else mels[i % 24] for i in range(240)]
want = [dec1.generate(enc1(m)).cpu() for m in batches[:24]]
torch.cuda.synchronize()
torch.cuda.empty_cache()          # torch's caching allocator keeps per-stream pools: compare what the DRIVER has free
free_a = torch.cuda.mem_get_info()[0]
for rnd in range(3):
    pipe = w.WhisperPipeline(eb2, db2, cfg2, workers=4)
    t = time.time()
    got = pipe.transcribe(batches)
    for i, g in enumerate(got):
        ref = dec1.generate(enc1(batches[i])).cpu() if i >= 24 else want[i]
        assert torch.equal(g.cpu(), ref), (rnd, i)
    print(f"pipeline round {rnd}: 240 batches on 4 workers in {time.time() - t:.2f} s, all bitwise equal to one engine", flush=True)
    del pipe, got
import gc
gc.collect()
torch.cuda.synchronize()
torch.cuda.empty_cache()
assert abs(torch.cuda.mem_get_info()[0] - free_a) < 64 * 2**20, "device memory is leaking across pipelines"
print("pipeline soak ok")
