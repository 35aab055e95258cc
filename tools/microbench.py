#!/usr/bin/env python3
"""Kernel micro-benchmarks through the debug hooks (GPU box only): rotating operands so nothing is cache-resident
between launches that would not be in the real pipeline (24 decoder layers -> 24 distinct weight / KV sets)."""
import argparse
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import whisper_trtllm_amd as w  # noqa: E402

lib = w._lib.load()
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
ST = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def timeit(fn, n_rot, iters=5):
    """us per launch.  The n_rot launches are captured into one graph (as the decode step is) and replayed, so the
    number includes the in-graph kernel boundary but no host launch cost."""
    for i in range(n_rot):
        fn(i)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            for i in range(n_rot):
                fn(i)
    g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / (iters * n_rot)


def bench_dec_attn(B=8, H=16, S=1500, L=24):
    d = 64 * H
    q = torch.randn(B, d, device="cuda") * 0.3
    k = torch.randn(L, B, H, S, 64, device="cuda")
    v = torch.randn(L, B, H, S, 64, device="cuda")
    out = torch.empty(B, d, device="cuda")
    cnt = torch.zeros(B, H, dtype=torch.int32, device="cuda")
    mb = B * H * S * 64 * 4 * 2 / 1e6
    for ns in (1, 2, 3, 4, 6, 8):
        part = torch.empty(B, H, ns, 68, device="cuda")
        us = timeit(lambda i: lib.wt_dbg_decode_attention(P(q), P(k[i]), P(v[i]), P(part), P(cnt), P(out), B, H, S, S, ns, ST()), L)
        print(f"dec_attn S={S} n_split={ns}: {us:7.2f} us  {mb / us * 1e-3 * 1e3:7.1f} TB/s")
    for length in (32, 224, 447):
        for ns in (1, 2, 4):
            part = torch.empty(B, H, ns, 68, device="cuda")
            us = timeit(lambda i: lib.wt_dbg_decode_attention(P(q), P(k[i]), P(v[i]), P(part), P(cnt), P(out), B, H, S, length, ns, ST()), L)
            print(f"dec_attn len={length} (cap {S}) n_split={ns}: {us:7.2f} us")


def bench_self_attn_tail(H=16, S=448, L=24):
    """VERDICT r2 item 3 (self-attention fused into the q|k|v GEMV by a last-arriver ticket per head): the tail would run on 16
    workgroups (one per head), each streaming that head's K/V of ALL B rows.  Emulation: the stand-alone kernel at B = 1 puts
    exactly 16 workgroups on the chip, one (row, head) each -- the fused tail costs about B x its streaming time."""
    d = 64 * H
    for B in (1, 8):
        q = torch.randn(B, d, device="cuda") * 0.3
        k = torch.randn(L, B, H, S, 64, device="cuda")
        v = torch.randn(L, B, H, S, 64, device="cuda")
        out = torch.empty(B, d, device="cuda")
        cnt = torch.zeros(B, H, dtype=torch.int32, device="cuda")
        part = torch.empty(B, H, 1, 68, device="cuda")
        for length in (8, 32, 96, 224, 447):
            us = timeit(lambda i: lib.wt_dbg_decode_attention(P(q), P(k[i]), P(v[i]), P(part), P(cnt), P(out), B, H, S, length, 1, ST()), L)
            print(f"self-attention B={B} ({B * H} workgroups) len={length}: {us:6.2f} us")


def bench_dec_attn_f16(H=16, S=1500, L=24):
    """cross-attention over fp16 resident caches (fp16 decoder engines): batch 8 / 16, key splits 1 / 2 / 4; WT_ATTN_U_HALF = keys per
    stream and iteration (WT_TUNING=1)"""
    d = 64 * H
    for B in (8, 16):
        q = torch.randn(B, d, device="cuda") * 0.3
        k = (torch.randn(L, B, H, S, 64, device="cuda") * 0.5).half()
        v = (torch.randn(L, B, H, S, 64, device="cuda") * 0.5).half()
        out = torch.empty(B, d, device="cuda")
        cnt = torch.zeros(B, H, dtype=torch.int32, device="cuda")
        mb = B * H * S * 64 * 2 * 2 / 1e6
        for ns in (1, 2, 4):
            part = torch.empty(B, H, ns, 68, device="cuda")
            us = timeit(lambda i: lib.wt_dbg_decode_attention_f16(P(q), P(k[i]), P(v[i]), P(part), P(cnt), P(out), None, None, None, B, H, S, S, ns, ST()), L)
            print(f"dec_attn f16 B={B} S={S} n_split={ns}: {us:7.2f} us  {mb / us:6.2f} TB/s  ({mb / us / 8:.3f} of 8 TB/s)")


def bench_dec_attn_folded(B=8, H=16, S=1500, L=24):
    """cross-attention launch with the plain query vs the folded query (LayerNorm statistics finished in the kernel)"""
    d = H * 64
    q = torch.randn(B, d, device="cuda") * 0.3
    h1 = torch.randn(B, d, device="cuda")
    r, t = torch.randn(d, device="cuda"), torch.randn(d, device="cuda")
    k = torch.randn(L, B, H, S, 64, device="cuda") * 0.5
    v = torch.randn(L, B, H, S, 64, device="cuda") * 0.5
    part = torch.empty(B * H * 16 * 68, device="cuda")
    cnt = torch.zeros(B * H, dtype=torch.int32, device="cuda")
    out = torch.empty(B, d, device="cuda")
    mb = L and 2 * B * H * S * 64 * 4 / 1e6
    us = timeit(lambda i: lib.wt_dbg_decode_attention(P(q), P(k[i]), P(v[i]), P(part), P(cnt), P(out), B, H, S, S, 2, ST()), L)
    print(f"cross-attention plain  query: {us:6.2f} us  {mb / us:6.2f} TB/s")
    us = timeit(lambda i: lib.wt_dbg_decode_attention_folded(P(q), P(k[i]), P(v[i]), P(part), P(cnt), P(out), P(h1), P(r), P(t), B, H, S, S, 2, ST()), L)
    print(f"cross-attention folded query: {us:6.2f} us  {mb / us:6.2f} TB/s")


def bench_skinny(B=8, L=24):
    for (N, K, xmode) in ((1024, 1024, 0), (1024, 1024, 4), (1024, 1024, 1), (1024, 1024, 5), (3072, 1024, 5), (4096, 1024, 5), (1024, 4096, 4), (51864, 1024, 5), (3072, 1024, 1), (4096, 1024, 1), (1024, 4096, 0), (51864, 1024, 1)):
        n_rot = L if N < 50000 else 4
        W = torch.randn(n_rot, N, K, device="cuda") * 0.02
        X = torch.randn(B, K, device="cuda")
        g, be, bias = torch.ones(K, device="cuda"), torch.zeros(K, device="cuda"), torch.zeros(N, device="cuda")
        Y = torch.empty(B, N, device="cuda")
        us = timeit(lambda i: lib.wt_dbg_skinny(P(X), P(g), P(be), P(W[i]), P(bias), None, P(Y), B, N, K, xmode, 0, 1.0, ST()), n_rot)
        print(f"skinny N={N} K={K} xmode={xmode}: {us:7.2f} us  {N * K * 4 / us * 1e-3:7.1f} TB/s")


def bench_skinny16(L=24):
    """decode GEMVs at batch 16 (the K-split plan <16, 2, 8>)"""
    bench_skinny(B=16, L=L)


def bench_skinny_floor(B=8):
    """Fixed cost of a decode-step GEMV launch: tiny problem, and the real shapes with cache-resident weights."""
    for (N, K, xmode, n_rot) in ((64, 256, 0, 1), (1024, 1024, 0, 1), (1024, 1024, 1, 1), (4096, 1024, 1, 1), (1024, 4096, 0, 1), (4096, 1024, 1, 24)):
        W = torch.randn(n_rot, N, K, device="cuda") * 0.02
        X = torch.randn(B, K, device="cuda")
        g, be, bias = torch.ones(K, device="cuda"), torch.zeros(K, device="cuda"), torch.zeros(N, device="cuda")
        Y = torch.empty(B, N, device="cuda")
        us = timeit(lambda i: lib.wt_dbg_skinny(P(X), P(g), P(be), P(W[i % n_rot]), P(bias), None, P(Y), B, N, K, xmode, 0, 1.0, ST()), 24)
        print(f"skinny N={N} K={K} xmode={xmode} rotating {n_rot}: {us:7.2f} us")


def bench_skinny_resident(B=8):
    """How fast is a decode GEMV whose weights wait in the Infinity Cache (256 MB) but NOT in the XCD's L2 (4 MB)?  The same launch over
    n_rot rotating weight sets: 1 set = L2 + Infinity Cache resident, 4-12 sets = Infinity Cache only (each XCD's share of every set is
    evicted from its L2 between uses), 24 sets = from HBM (what the decode step sees today)."""
    for (N, K, xmode) in ((4096, 1024, 5), (1024, 4096, 4), (3072, 1024, 5), (1024, 1024, 4)):
        for n_rot in (1, 2, 4, 8, 12, 24, 48):
            if n_rot * N * K * 4 > 1.2e9:
                continue
            W = torch.randn(n_rot, N, K, device="cuda") * 0.02
            X = torch.randn(B, K, device="cuda")
            g, be, bias = torch.ones(K, device="cuda"), torch.zeros(K, device="cuda"), torch.zeros(N, device="cuda")
            Y = torch.empty(B, N, device="cuda")
            us = timeit(lambda i: lib.wt_dbg_skinny(P(X), P(g), P(be), P(W[i % n_rot]), P(bias), None, P(Y), B, N, K, xmode, 0, 1.0, ST()), 48)
            print(f"skinny N={N} K={K} xmode={xmode} rotating {n_rot:2d} sets ({n_rot * N * K * 4 / 1e6:6.1f} MB): {us:7.2f} us")


def bench_fold_fc1(B=8, L=24, d=1024, F=4096):
    """fold the cross out-projection through LN3 into fc1 (6 launches per decoder layer instead of 7).
    Today: cross_out (d x d, split-merged activations) + fc1 (LN + F x d + GELU) + fc2 (d x F).  Folded: ONE pair launch computing
    h2 = h1 + Wco.ctx (d x d) and u = [G.Wco | G].[ctx ; h1] (F x 2d), then fc2 finishing x = gelu((u - mean.r).rstd + t) on its own
    activation slice (every workgroup re-does the GELU of all B x F values it multiplies: 128 per lane).  Timing of both sides."""
    X = torch.randn(B, d, device="cuda")
    X2 = torch.randn(B, d, device="cuda")
    g, be = torch.ones(d, device="cuda"), torch.zeros(d, device="cuda")
    Wco = torch.randn(L, d, d, device="cuda") * 0.02
    Wfc1 = torch.randn(L, F, d, device="cuda") * 0.02
    Wfold = torch.randn(L, F, 2 * d, device="cuda") * 0.02
    Wfc2 = torch.randn(L, d, F, device="cuda") * 0.02
    bd, bF = torch.zeros(d, device="cuda"), torch.zeros(F, device="cuda")
    Yd, YF = torch.empty(B, d, device="cuda"), torch.empty(B, F, device="cuda")
    U = torch.randn(B, F, device="cuda")
    r, t = torch.randn(F, device="cuda"), torch.randn(F, device="cuda")
    us_co = timeit(lambda i: lib.wt_dbg_skinny(P(X), P(g), P(be), P(Wco[i]), P(bd), P(X2), P(Yd), B, d, d, 4, 0, 1.0, ST()), L)
    us_fc1 = timeit(lambda i: lib.wt_dbg_skinny(P(X), P(g), P(be), P(Wfc1[i]), P(bF), None, P(YF), B, F, d, 5, 1, 1.0, ST()), L)
    us_fc2 = timeit(lambda i: lib.wt_dbg_skinny(P(U), P(g), P(be), P(Wfc2[i]), P(bd), P(X2), P(Yd), B, d, F, 4, 0, 1.0, ST()), L)
    us_pair = timeit(lambda i: lib.wt_dbg_skinny_pair(P(X), P(Wco[i]), P(bd), P(X2), P(Yd), d, d, P(X), P(X2), P(Wfold[i]), P(bF), P(YF), F, 2 * d, B, ST()), L)
    us_fc2g = timeit(lambda i: lib.wt_dbg_skinny_gelu_in(P(U), P(r), P(t), P(Wfc2[i]), P(bd), P(X2), P(Yd), B, d, F, ST()), L)
    print(f"today : cross_out {us_co:6.2f} + fc1 {us_fc1:6.2f} + fc2 {us_fc2:6.2f} = {us_co + us_fc1 + us_fc2:6.2f} us per layer")
    print(f"folded: pair [d x d | F x 2d] {us_pair:6.2f} + fc2 with the GELU / LN finish in its prologue {us_fc2g:6.2f} = {us_pair + us_fc2g:6.2f} us per layer"
          f"  (the prologue still lacks the LayerNorm statistics of the B rows: a lower bound)")


def bench_gemm(M=12000):
    for (N, K, act) in ((3072, 1024, 0), (1024, 1024, 0), (4096, 1024, 1), (1024, 4096, 0), (2048, 1024, 0)):
        A = torch.randn(M, K, device="cuda")
        W = torch.randn(4, N, K, device="cuda") * 0.03
        bias = torch.zeros(N, device="cuda")
        C = torch.empty(M, N, device="cuda")
        us = timeit(lambda i: lib.wt_dbg_gemm(P(A), K, P(W[i]), P(bias), None, P(C), M, N, K, act, ST()), 4, iters=3)
        print(f"gemm M={M} N={N} K={K} act={act}: {us:8.1f} us  {2.0 * M * N * K / us * 1e-6:6.1f} TFLOP/s")


def bench_gemm_x3(M=12000):
    """launch_gemm_x3 (fp32 product from six bf16 MFMAs of exactly split operands) beside the native fp32 MFMA kernel on the encoder's
    shapes; TFLOP/s of USEFUL fp32 work (2 M N K); the split kernel issues 6x that in bf16 MFMA flops."""
    for (N, K, act) in ((3072, 1024, 0), (1024, 1024, 0), (4096, 1024, 1), (1024, 4096, 0), (2048, 1024, 0)):
        zero = 0.0 if os.environ.get("MB_ZERO") else 1.0     # all-zero operands: the clock the chip holds when the MFMAs toggle nothing
        A = torch.randn(M, K, device="cuda") * zero
        W = torch.randn(4, N, K, device="cuda") * 0.03 * zero
        bias = torch.zeros(N, device="cuda")
        C = torch.empty(M, N, device="cuda")
        a_pl = torch.empty(3 * M * K, dtype=torch.bfloat16, device="cuda")
        w_pl = torch.empty(4, 3 * N * K, dtype=torch.bfloat16, device="cuda")
        for i in range(4):
            lib.wt_dbg_gemm_x3(P(A), P(W[i]), P(bias), None, P(C), M, N, K, act, P(a_pl), P(w_pl[i]), 0, ST())    # leaves the planes in place
        us_n = timeit(lambda i: lib.wt_dbg_gemm(P(A), K, P(W[i]), P(bias), None, P(C), M, N, K, act, ST()), 4, iters=5)
        us_h = timeit(lambda i: lib.wt_dbg_gemm_x3(P(A), P(W[i]), P(bias), None, P(C), M, N, K, act, P(a_pl), P(w_pl[i]), 2, ST()), 4, iters=5)
        tf = lambda u: 2.0 * M * N * K / u * 1e-6
        print(f"M={M} N={N} K={K} act={act}: native fp32 MFMA {us_n:8.1f} us {tf(us_n):6.1f} TF | x3 (planes in place) {us_h:8.1f} us {tf(us_h):6.1f} TF useful = {6 * tf(us_h):6.0f} TF of bf16 MFMA")


def bench_gemm_shapes():
    """fp32 GEMM on the encoder shapes of medium.en / small.en (batch 8) and tiny.en (batch 1); run once per WT_GEMM_BN setting"""
    for (M, N, K) in ((12000, 1024, 1024), (12000, 1024, 4096), (12000, 2048, 1024), (12000, 3072, 1024), (12000, 4096, 1024),
                      (12000, 768, 768), (12000, 768, 3072), (12000, 2304, 768), (12000, 3072, 768), (12000, 1536, 768),
                      (1500, 384, 384), (1500, 1152, 384), (1500, 1536, 384), (1500, 384, 1536)):
        A = torch.randn(M, K, device="cuda")
        W = torch.randn(4, N, K, device="cuda") * 0.03
        bias = torch.zeros(N, device="cuda")
        C = torch.empty(M, N, device="cuda")
        us = timeit(lambda i: lib.wt_dbg_gemm(P(A), K, P(W[i]), P(bias), None, P(C), M, N, K, 0, ST()), 4, iters=10)
        print(f"gemm M={M} N={N} K={K}: {us:8.1f} us  {2.0 * M * N * K / us * 1e-6:6.1f} TFLOP/s")


def bench_gemm_rounds():
    """fp32 GEMM: how much of the distance to the MFMA peak is per-launch (ramp + uneven finish of the co-resident workgroups)?  Same
    N, K with M = exactly 1, 2, 4, 8, 16 rounds of the chip's 768 workgroup slots (M = 12000 is the encoder's 0.98 / 2.9 / 3.9 rounds)."""
    for (N, K) in ((1024, 1024), (1024, 4096), (4096, 1024)):
        for rounds in (1, 2, 4, 8, 16):
            M = rounds * 768 * 128 * 128 // N
            if M * max(N, K) > (1 << 29):
                continue
            A = torch.randn(M, K, device="cuda")
            W = torch.randn(4, N, K, device="cuda") * 0.03
            bias = torch.zeros(N, device="cuda")
            C = torch.empty(M, N, device="cuda")
            us = timeit(lambda i: lib.wt_dbg_gemm(P(A), K, P(W[i]), P(bias), None, P(C), M, N, K, 0, ST()), 4, iters=5)
            print(f"gemm M={M} N={N} K={K} ({rounds} rounds of 768 tiles): {us:8.1f} us  {2.0 * M * N * K / us * 1e-6:6.1f} TFLOP/s")


def bench_gemm_steady():
    """fp32 GEMM at 8 rounds of the chip (per-launch effects amortised): the steady-state rate, for the WT_GEMM_ABLATE probes"""
    for (M, N, K) in ((98304, 1024, 1024), (24576, 1024, 4096)):
        A = torch.randn(M, K, device="cuda")
        W = torch.randn(4, N, K, device="cuda") * 0.03
        bias = torch.zeros(N, device="cuda")
        C = torch.empty(M, N, device="cuda")
        us = timeit(lambda i: lib.wt_dbg_gemm(P(A), K, P(W[i]), P(bias), None, P(C), M, N, K, 0, ST()), 4, iters=5)
        print(f"gemm M={M} N={N} K={K}: {us:8.1f} us  {2.0 * M * N * K / us * 1e-6:6.1f} TFLOP/s (if it were the full product)")


def bench_gemm_fixed(M=12000, N=1024):
    """fixed per-tile cost of the fp32 GEMM: K sweep at constant output size (752 tiles = one round)"""
    for K in (16, 64, 256, 512, 1024, 2048, 4096):
        A = torch.randn(M, K, device="cuda")
        W = torch.randn(4, N, K, device="cuda") * 0.03
        bias = torch.zeros(N, device="cuda")
        C = torch.empty(M, N, device="cuda")
        us = timeit(lambda i: lib.wt_dbg_gemm(P(A), K, P(W[i]), P(bias), None, P(C), M, N, K, 0, ST()), 4, iters=20)
        print(f"gemm M={M} N={N} K={K}: {us:8.1f} us  {2.0 * M * N * K / us * 1e-6:6.1f} TFLOP/s")


def bench_gemm_f16(M=None):
    for m in ([M] if M else [12000, 24000]):
        _bench_gemm_f16(m)


def _bench_gemm_f16(M):
    for (N, K, act, oh) in ((3072, 1024, 0, 0), (1024, 1024, 0, 0), (4096, 1024, 1, 1), (1024, 4096, 0, 0), (1024, 3072, 1, 0)):
        A = torch.randn(M, K, device="cuda").half()
        W = (torch.randn(4, N, K, device="cuda") * 0.03).half()
        bias = torch.zeros(N, device="cuda")
        C = torch.empty(M, N, device="cuda", dtype=torch.float16 if oh else torch.float32)
        us = timeit(lambda i: lib.wt_dbg_gemm_f16(P(A), K, P(W[i]), P(bias), None, P(C), M, N, K, act, oh, ST()), 4, iters=3)
        print(f"gemm_f16 M={M} N={N} K={K} act={act} out_half={oh}: {us:8.1f} us  {2.0 * M * N * K / us * 1e-6:6.1f} TFLOP/s")
        if not oh and not act:   # as the engine runs out-proj / fc2: fp32 residual stream read and written in place
            us = timeit(lambda i: lib.wt_dbg_gemm_f16(P(A), K, P(W[i]), P(bias), P(C), P(C), M, N, K, act, oh, ST()), 4, iters=3)
            print(f"gemm_f16 M={M} N={N} K={K} act={act} out_half={oh} +resid: {us:8.1f} us  {2.0 * M * N * K / us * 1e-6:6.1f} TFLOP/s")


def bench_gemm_f16_variants():
    """fp16 GEMM kernels side by side on the encoder's shapes: 2 = 128x128x64 two-stage, 3 = persistent 256x128x64, 4 = persistent
    256x256x32 (round 3); 0 = what the shape rule picks"""
    for M in (12000, 24000):
        for (N, K, act, oh) in ((3072, 1024, 0, 1), (4096, 1024, 1, 1), (1024, 4096, 0, 0), (1024, 1024, 0, 0), (2048, 1024, 0, 1)):
            A = torch.randn(M, K, device="cuda").half()
            W = (torch.randn(4, N, K, device="cuda") * 0.03).half()
            bias = torch.zeros(N, device="cuda")
            C = torch.empty(M, N, device="cuda", dtype=torch.float16 if oh else torch.float32)
            row = []
            for variant in (0, 2, 3, 4, 5):
                fn = (lambda i: lib.wt_dbg_gemm_f16(P(A), K, P(W[i]), P(bias), None, P(C), M, N, K, act, oh, ST())) if variant == 0 else \
                     (lambda i: lib.wt_dbg_gemm_f16_variant(P(A), K, P(W[i]), P(bias), None, P(C), M, N, K, act, oh, variant, ST()))
                us = timeit(fn, 4, iters=5)
                row.append(f"v{variant} {us:7.1f} us {2.0 * M * N * K / us * 1e-6:6.0f} TF")
            print(f"gemm_f16 M={M} N={N} K={K} act={act} out_half={oh}: " + " | ".join(row))


def bench_gemm_vs_library():
    """Yardstick, not a product path: the vendor library (torch.nn.functional.linear -> hipBLASLt / rocBLAS) on the encoder's GEMM
    shapes beside the engine's own kernels.  The library rows are the PLAIN product (bias only, no GELU / residual / fp16-cast
    epilogue work beyond what F.linear does); the engine rows run the epilogue the encoder uses on that shape."""
    import torch.nn.functional as F
    for dt, Ms in ((torch.float32, (12000,)), (torch.float16, (12000, 24000))):
        for M in Ms:
            for (N, K) in ((3072, 1024), (1024, 1024), (4096, 1024), (1024, 4096)):
                A = torch.randn(M, K, device="cuda").to(dt)
                W = (torch.randn(4, N, K, device="cuda") * 0.03).to(dt)
                bias = torch.zeros(N, device="cuda")
                bias_l = bias.to(dt)
                out_l = torch.empty(M, N, device="cuda", dtype=dt)
                us_lib = timeit(lambda i: torch.addmm(bias_l, A, W[i].t(), out=out_l), 4, iters=5)
                if dt == torch.float32:
                    C = torch.empty(M, N, device="cuda")
                    us = timeit(lambda i: lib.wt_dbg_gemm(P(A), K, P(W[i]), P(bias), None, P(C), M, N, K, 0, ST()), 4, iters=5)
                else:
                    oh = 1 if N >= 2048 else 0
                    C = torch.empty(M, N, device="cuda", dtype=torch.float16 if oh else torch.float32)
                    us = timeit(lambda i: lib.wt_dbg_gemm_f16(P(A), K, P(W[i]), P(bias), None, P(C), M, N, K, 0, oh, ST()), 4, iters=5)
                tf = lambda u: 2.0 * M * N * K / u * 1e-6
                print(f"{str(dt)[6:]:8s} M={M} N={N} K={K}: engine {us:7.1f} us {tf(us):7.1f} TF | library {us_lib:7.1f} us {tf(us_lib):7.1f} TF")


def bench_gemm_f16_v4(M=24000):
    """the 256x256x32 kernel alone on its two shapes (for the WT_HGEMM_ABLATE probe: 1 no MFMA, 2 no LDS-DMA, 3 no fragment reads)"""
    for (N, K, act, oh) in ((3072, 1024, 0, 1), (4096, 1024, 0, 1), (3072, 4096, 0, 1)):
        A = torch.randn(M, K, device="cuda").half()
        W = (torch.randn(4, N, K, device="cuda") * 0.03).half()
        bias = torch.zeros(N, device="cuda")
        C = torch.empty(M, N, device="cuda", dtype=torch.float16)
        us = timeit(lambda i: lib.wt_dbg_gemm_f16_variant(P(A), K, P(W[i]), P(bias), None, P(C), M, N, K, act, oh, 4, ST()), 4, iters=5)
        print(f"v4 M={M} N={N} K={K}: {us:7.1f} us  ({2.0 * M * N * K / us * 1e-6:6.0f} TF if it were the full product)")


def bench_gemm_f16_resident(M=24000, N=1024):
    """Is the fp16 GEMM's K loop bound by the latency of its compulsory L2 misses?  Same launches with lda = 0: every row of A is the
    same 2*K bytes, so A is L1/L2-resident and only W (2-8 MB) streams -- if the loop speeds up a lot, it was waiting for misses."""
    for variant in (2, 3):
        for K in (1024, 4096):
            A = torch.randn(M, K, device="cuda").half()
            W = (torch.randn(4, N, K, device="cuda") * 0.03).half()
            bias = torch.zeros(N, device="cuda")
            C = torch.empty(M, N, device="cuda", dtype=torch.float16)
            for lda, label in ((K, "normal A"), (0, "A resident (lda = 0)")):
                us = timeit(lambda i: lib.wt_dbg_gemm_f16_variant(P(A), lda, P(W[i]), P(bias), None, P(C), M, N, K, 0, 1, variant, ST()), 4, iters=5)
                print(f"variant {variant} M={M} N={N} K={K} {label:22s}: {us:8.1f} us  {2.0 * M * N * K / us * 1e-6:7.1f} TFLOP/s")


def bench_logmel(B=8):
    fe = w.audio.LogMelFrontend()
    wav = torch.randn(B, 480000, device="cuda") * 0.1
    us = timeit(lambda i: fe(wav), 1, iters=5)
    print(f"log-mel front-end, {B} x 30 s on the GPU: {us / 1e3:8.3f} ms  ({B * 30 / (us * 1e-6):.0f} audio-s/s)")


def bench_enc_attn(B=8, S=1500, H=16):
    qkv = torch.randn(B * S, 3 * 64 * H, device="cuda")
    ctx = torch.empty(B * S, 64 * H, device="cuda")
    us = timeit(lambda i: lib.wt_dbg_encoder_attention(P(qkv), P(ctx), B, S, H, ST()), 1, iters=5)
    print(f"enc_attn B={B} S={S} H={H}: {us:8.1f} us  {4.0 * B * H * S * S * 64 / us * 1e-6:6.1f} TFLOP/s")
    planes = torch.empty(3, B * S, 64 * H, dtype=torch.bfloat16, device="cuda")
    us = timeit(lambda i: lib.wt_dbg_encoder_attention_split(P(qkv), P(planes), B, S, H, ST()), 1, iters=5)
    print(f"enc_attn (context out as three bf16 planes) B={B} S={S} H={H}: {us:8.1f} us  {4.0 * B * H * S * S * 64 / us * 1e-6:6.1f} TFLOP/s")
    qkv_pl = torch.empty(3 * B * S * 3 * 64 * H, dtype=torch.bfloat16, device="cuda")
    lib.wt_dbg_encoder_attention_x3(P(qkv), P(qkv_pl), P(planes), B, S, H, 0, ST())
    us = timeit(lambda i: lib.wt_dbg_encoder_attention_x3(P(qkv), P(qkv_pl), P(planes), B, S, H, 1, ST()), 1, iters=5)
    print(f"enc_attn_x3 (q|k|v planes in, context planes out) B={B} S={S} H={H}: {us:8.1f} us  {4.0 * B * H * S * S * 64 / us * 1e-6:6.1f} TFLOP/s useful")
    qkv_h, ctx_h = qkv.half(), ctx.half()
    us = timeit(lambda i: lib.wt_dbg_encoder_attention_f16(P(qkv_h), P(ctx_h), B, S, H, ST()), 1, iters=5)
    print(f"enc_attn_f16 B={B} S={S} H={H}: {us:8.1f} us  {4.0 * B * H * S * S * 64 / us * 1e-6:6.1f} TFLOP/s")


def bench_dec_attn_resident(B=8, H=16, S=1500):
    """Same cross-attention launch with the K/V of ONE layer re-read every launch: 98 MB stays in the 256 MB
    Infinity Cache, i.e. the rate a prefetch-ahead design could reach."""
    for L in (1, 2, 4):
        d = 64 * H
        q = torch.randn(B, d, device="cuda") * 0.3
        k = torch.randn(L, B, H, S, 64, device="cuda")
        v = torch.randn(L, B, H, S, 64, device="cuda")
        out = torch.empty(B, d, device="cuda")
        cnt = torch.zeros(B, H, dtype=torch.int32, device="cuda")
        part = torch.empty(B, H, 2, 68, device="cuda")
        us = timeit(lambda i: lib.wt_dbg_decode_attention(P(q), P(k[i % L]), P(v[i % L]), P(part), P(cnt), P(out), B, H, S, S, 2, ST()), 24)
        print(f"dec_attn S={S} n_split=2, {L} rotating layer(s) ({L * 98} MB working set): {us:7.2f} us")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("what", nargs="*", default=["dec_attn", "skinny", "gemm", "enc_attn"])
    a = ap.parse_args()
    for name in a.what:
        globals()["bench_" + name]()

