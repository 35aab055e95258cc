#!/usr/bin/env python3
"""Wall-clock breakdown of one pass (encoder / begin = cross-KV projection / decode steps / readback) on the GPU box."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import whisper_trtllm_amd as w  # noqa: E402

model = sys.argv[1] if len(sys.argv) > 1 else "whisper-medium.en"
cfg = w.synthetic.get_config(model)
weights = w.synthetic.make_weights(cfg, 0)
enc = w.WhisperEncoderEngine(w.convert.build_encoder_engine(cfg, weights))
dec = w.WhisperDecoderEngine(w.convert.build_decoder_engine(cfg, weights), cfg)
mel = torch.from_numpy(w.synthetic.make_mel(cfg, 0, 8)).cuda()


def t(fn, n=3):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


hidden = enc(mel)
print("encoder            %.2f ms" % t(lambda: enc(mel)))
print("begin (cross-KV)   %.2f ms" % t(lambda: dec.begin(hidden)))
dec.begin(hidden)
print("33 steps + 3 polls %.2f ms" % t(lambda: (dec.begin(hidden, force_eos_step=32), dec.steps(16), dec.poll(), dec.steps(16), dec.poll(), dec.steps(16), dec.poll())))
print("generate(eos@32)   %.2f ms" % t(lambda: dec.generate(hidden, force_eos_step=32)))
print("enc + generate     %.2f ms" % t(lambda: dec.generate(enc(mel), force_eos_step=32)))
for chunk in (4, 8, 16, 64, 447):
    print("generate(447 steps, chunk=%3d) %.2f ms" % (chunk, t(lambda: dec.generate(hidden, chunk=chunk), 2)))
dec.begin(hidden)
t0 = time.perf_counter()
dec.steps(8)
t1 = time.perf_counter()
torch.cuda.synchronize()
print("host time to enqueue 8 step graphs: %.3f ms" % ((t1 - t0) * 1e3))
t0 = time.perf_counter()
dec.poll()
print("poll on an idle stream: %.3f ms" % ((time.perf_counter() - t0) * 1e3))
