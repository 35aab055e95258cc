import sys, time; sys.path.insert(0,'.')
import numpy as np, torch
import whisper_trtllm_amd as wt
fe = wt.audio.LogMelFrontend()
x = torch.randn(8, 480000, device="cuda") * 0.1
for _ in range(3): fe(x)
torch.cuda.synchronize(); t=time.perf_counter()
for _ in range(20): fe(x)
torch.cuda.synchronize(); print("front-end 8 x 30 s: %.3f ms" % ((time.perf_counter()-t)/20*1e3))
