"""fp32 GEMM K loop with exactly 1, 2 or 3 workgroups per CU (256, 512, 768 tiles of 128x128, K = 4096): how close does ONE wave per SIMD
get to the MFMA rate?  (Round 2: 117-119 TFLOP/s with 1 workgroup per CU, 130-132 with 2 or 3; a software-pipelined loop that kept
one wave's MFMA queue full over the barrier reached 123 at 1 per CU and LOST 3 % at 2-3 per CU -- DESIGN.md section 9.2.)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import whisper_trtllm_amd as w

lib = w._lib.load()
P = lambda t: t.data_ptr() if t is not None else None
ST = lambda: torch.cuda.current_stream().cuda_stream
K = 4096
for tiles_m, tiles_n in ((16, 16), (32, 16), (48, 16)):
    M, N = 128 * tiles_m, 128 * tiles_n
    A = torch.randn(M, K, device="cuda")
    W = torch.randn(N, K, device="cuda") * 0.03
    C = torch.zeros(M, N, device="cuda")
    for _ in range(50):
        lib.wt_dbg_gemm(P(A), K, P(W), None, None, P(C), M, N, K, 0, ST())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        lib.wt_dbg_gemm(P(A), K, P(W), None, None, P(C), M, N, K, 0, ST())
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 50
    print(f"tiles={tiles_m * tiles_n} ({tiles_m * tiles_n // 256}/CU) K={K}: {us:8.1f} us  {2.0 * M * N * K / us * 1e-6:6.1f} TFLOP/s")
