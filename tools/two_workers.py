#!/usr/bin/env python3
"""N workers per GPU (runtime.WhisperPipeline: N engine pairs, N host threads, N HIP streams) on whole passes of the headline workload
(encoder + 447-step greedy decode of 8 utterances): audio-s/s for N = 1 .. 4.  The decode is launch-latency bound, so a second chain
fills its gaps.  (Two PROCESSES sharing the GPU: `python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2 --share-gpu
--dist-backend gloo`.)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import whisper_trtllm_amd as w

model = sys.argv[1] if len(sys.argv) > 1 else "whisper-medium.en"
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 6
B = int(sys.argv[3]) if len(sys.argv) > 3 else 8
cfg = w.synthetic.get_config(model)
weights = w.synthetic.make_weights(cfg, 0)
eb, db = w.convert.build_encoder_engine(cfg, weights), w.convert.build_decoder_engine(cfg, weights)
mels = [torch.from_numpy(w.synthetic.make_mel(cfg, index=B * i, batch=B)).cuda() for i in range(passes)]
for n in ([int(x) for x in sys.argv[4].split(',')] if len(sys.argv) > 4 else (1, 2, 3, 4)):
    pipe = w.WhisperPipeline(eb, db, cfg, workers=n)
    pipe.transcribe(mels[:n])
    torch.cuda.synchronize()
    t = time.perf_counter()
    pipe.transcribe(mels)
    torch.cuda.synchronize()
    el = time.perf_counter() - t
    print(f"{model}: {n} worker(s), {passes} passes of {B} x 30 s: {el:.3f} s  ({passes * 30 * B / el:.1f} audio-s/s)", flush=True)
    del pipe
