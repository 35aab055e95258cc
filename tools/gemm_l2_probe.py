"""L2 behaviour of the fp32 encoder GEMM: run under `rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum` (own pass) and
`--pmc FETCH_SIZE`; each shape once with lda = K and once with the A rows padded by 32 floats (is it a set-conflict effect of
power-of-two row strides?).  Launch order: see SHAPES; tools/pmc_table.py prints per-kernel means, so every (shape, lda) pair is
launched with a distinct workgroup count only by shape -- read the per-launch rows of the CSV for the lda split."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import whisper_trtllm_amd as w

lib = w._lib.load()
P = lambda t: t.data_ptr() if t is not None else None
ST = lambda: torch.cuda.current_stream().cuda_stream
SHAPES = [(12000, 1024, 1024), (12000, 1024, 4096), (12000, 4096, 1024), (12000, 3072, 1024)]
for (M, N, K) in SHAPES:
    for pad in (0, 32):
        A = torch.randn(M, K + pad, device="cuda")
        W = torch.randn(N, K, device="cuda") * 0.03
        bias = torch.zeros(N, device="cuda")
        C = torch.empty(M, N, device="cuda")
        for _ in range(3):
            lib.wt_dbg_gemm(P(A), K + pad, P(W), P(bias), None, P(C), M, N, K, 0, ST())
        torch.cuda.synchronize()
print("done")
