# Manual GPU check (imports the oracle, hence under tests/): python tests/manual/headline_parity.py
import sys, os, time
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "oracle")]
import numpy as np, torch
import whisper_trtllm_amd as w
import cpu_ref
cfg = w.synthetic.get_config("whisper-medium.en")
weights = w.synthetic.make_weights(cfg, 77)
B, steps = 8, 447
mel = w.synthetic.make_mel(cfg, index=300, batch=B)
enc = w.WhisperEncoderEngine(w.convert.build_encoder_engine(cfg, weights))
dec = w.WhisperDecoderEngine(w.convert.build_decoder_engine(cfg, weights), cfg)
hidden = enc(torch.from_numpy(mel).cuda())
V = cfg["vocab_size"]
trace = torch.zeros(B, steps, V, dtype=torch.float32, device="cuda")
ids = dec.generate(hidden, logits_trace=trace).cpu()
torch.set_num_threads(16)
W = cpu_ref.to_torch(weights)
t0 = time.time()
with torch.no_grad():
    h = cpu_ref.encoder_forward(W, cfg, torch.from_numpy(mel))
    print("oracle encoder", time.time() - t0, flush=True)
    ids_ref, logits_ref = cpu_ref.greedy_search(W, cfg, h, max_length=steps + 1, return_logits=True)
print("oracle total", time.time() - t0, flush=True)
top2 = torch.topk(logits_ref[:, 1:], 2, dim=-1).values
m = (top2[..., 0] - top2[..., 1])
print("min margin", m.min().item(), "per row", m.min(1).values.tolist())
print("enc err", (hidden.cpu() - h).abs().max().item(), "scale", h.abs().max().item())
err = (trace.cpu() - logits_ref).abs().amax(dim=(0, 2))
print("max logits err", err.max().item(), "at step", int(err.argmax()))
print("ids equal", bool(torch.equal(ids, ids_ref)), "first mismatch", (ids != ids_ref).nonzero()[:3].tolist())
