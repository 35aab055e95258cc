#!/usr/bin/env python3
"""Soak of the continuous mode (GPU box): N utterances (default 50,000) with random forced lengths through ONE open stream of 8 slots --
more than 65,536 decoder steps and more than 32,768 admissions, i.e. past the wrap of every counter the mailbox word carries (16-bit steps
retired, 15-bit admissions) -- every id row compared with the batch decode of the same utterance.  whisper-tiny.en, synthetic weights."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import whisper_trtllm_amd as w

N = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
K, SLOTS, ML = 32, 8, 32
LENGTHS = (3, 6, 9, 14, 20)
cfg = w.synthetic.get_config("whisper-tiny.en")
weights = w.synthetic.make_weights(cfg, 2)
enc = w.WhisperEncoderEngine(w.convert.build_encoder_engine(cfg, weights))
dec = w.WhisperDecoderEngine(w.convert.build_decoder_engine(cfg, weights), cfg)
hidden = torch.cat([enc(torch.from_numpy(w.synthetic.make_mel(cfg, index=16 * i, batch=16)).cuda()) for i in range(K // 16)])
eos = int(cfg["eos_token_id"])
want = {}
for fe in LENGTHS:
    for a in range(0, K, 16):
        ids = dec.generate(hidden[a:a + 16], max_length=ML, force_eos_steps=[fe] * 16).cpu().numpy()
        for j, row in enumerate(ids):
            cut = np.flatnonzero(row[1:] == eos)
            want[(a + j, fe)] = row[: cut[0] + 2] if len(cut) else row
rng = np.random.default_rng(0)
ks, fes = rng.integers(0, K, N), rng.choice(LENGTHS, N)
st = dec.stream(slots=SLOTS, pool_rows=64, max_length=ML)
index_of, nxt, done, waiting = {}, 0, 0, 0
t0 = time.time()
while done < N:
    while nxt < N and waiting < 16 and st.free_rows() >= 1:
        n = int(min(16, N - nxt, st.free_rows()))
        sel = torch.from_numpy(ks[nxt:nxt + n]).cuda()
        for j, h in enumerate(st.submit(hidden.index_select(0, sel), [int(v) for v in fes[nxt:nxt + n]])):
            index_of[h] = nxt + j
        nxt += n
        waiting += n
    more = nxt < N
    _, waiting = st.run(min_waiting=(max(1, waiting) if st.free_rows() == 0 else SLOTS) if more else 0)
    for h, ids in st.collect():
        i = index_of.pop(h)
        ref = want[(int(ks[i]), int(fes[i]))]
        assert len(ids) == len(ref) and (ids == ref).all(), (i, ids, ref)
        done += 1
    if done and done % 10000 < 16:
        print(f"{done} utterances, {st.n_steps} steps, {time.time() - t0:.1f} s", flush=True)
el = time.time() - t0
util = float(sum(len(want[(int(k), int(f))]) - 1 for k, f in zip(ks, fes))) / (st.n_steps * SLOTS)
print(f"stream soak ok: {N} utterances, {st.n_steps} decoder steps (16-bit step counter wrapped {st.n_steps // 65536}x, 15-bit admission counter "
      f"{N // 32768}x), slot utilisation {util:.3f}, {el:.1f} s, every id row equal to the batch decode")
