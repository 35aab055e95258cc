"""Kernel-level GPU numerics: each hand-written HIP kernel (through the debug hooks of the C-ABI library)
against a plain fp32 torch reference of the same op, at edge shapes.  Pattern follows the reference's op
tests (tests/functional/test_matmul.py:58-67 fp32 1e-5-ish, test_layer_norm.py:75, test_layer.py:541-767)."""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    import whisper_trtllm_amd as w
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU is visible")
    return w._lib.load()


def P(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).float()


@pytest.mark.parametrize("M,N,K,act,use_res", [
    (128, 128, 32, 0, False), (200, 136, 240, 1, False), (1500, 384, 384, 0, True), (97, 1000, 1536, 1, True),
    (3000, 128, 240, 1, False), (33, 51, 4, 0, False), (1024, 3072, 1024, 0, False)])
def test_gemm(lib, M, N, K, act, use_res):
    A, W, b = _rand(M, K, seed=1), _rand(N, K, seed=2, scale=K ** -0.5), _rand(N, seed=3)
    R = _rand(M, N, seed=4) if use_res else None
    ref = F.linear(A.double(), W.double(), b.double())
    if act:
        ref = F.gelu(ref)
    if use_res:
        ref = ref + R.double()
    Ad, Wd, bd = A.cuda(), W.cuda(), b.cuda()
    C = R.cuda().clone() if use_res else torch.empty(M, N, device="cuda")
    assert lib.wt_dbg_gemm(P(Ad), K, P(Wd), P(bd), P(C) if use_res else None, P(C), M, N, K, act, _stream()) == 0
    torch.cuda.synchronize()
    err = (C.cpu().double() - ref).abs().max().item()
    assert err < 2e-5 * max(1.0, ref.abs().max().item()), err


@pytest.mark.parametrize("M,N,K,act,use_res,out_split", [
    (128, 128, 16, 0, False, False), (128, 128, 32, 0, False, False), (200, 136, 240, 1, False, False), (1500, 384, 384, 0, True, False),
    (97, 1000, 1536, 1, True, False), (1024, 3072, 1024, 0, False, False), (1500, 1536, 384, 1, False, True), (333, 200, 48, 1, False, True),
    (4100, 3210, 64, 1, False, True), (12000, 1024, 4096, 0, True, False)])
def test_gemm_x3_is_an_fp32_gemm(lib, M, N, K, act, use_res, out_split):
    """launch_gemm_x3: the fp32 product formed on the bf16 matrix cores from exactly split operands (x = b1 + b2 + b3, six of the nine
    partial products).  It must be AS ACCURATE AS the native fp32 MFMA kernel: both against fp64 on the same inputs, the split kernel's
    error within 1.5x of the native kernel's (measured: equal or smaller), and inside the native test's absolute bar.  One to 256 K
    steps (fewer than, as many as and more than its three pipeline stages), ragged tiles, every epilogue it is used with (plain, GELU,
    residual in place, three-plane bf16 output)."""
    A, W, b = _rand(M, K, seed=1), _rand(N, K, seed=2, scale=K ** -0.5), _rand(N, seed=3)
    # a few rows with a wide dynamic range and heavy cancellation: where dropping low-order bits would show first
    A[:3] *= torch.tensor([1e3, 1e-3, 1.0]).view(3, 1)
    A[3, 1::2] = -A[3, 0::2][: A[3, 1::2].numel()] * (1 + 1e-4)
    R = _rand(M, N, seed=4) if use_res else None
    ref = F.linear(A.double(), W.double(), b.double())
    if act:
        ref = F.gelu(ref)
    if use_res:
        ref = ref + R.double()
    Ad, Wd, bd = A.cuda(), W.cuda(), b.cuda()
    a_pl = torch.empty(3 * M * K, dtype=torch.bfloat16, device="cuda")
    w_pl = torch.empty(3 * N * K, dtype=torch.bfloat16, device="cuda")
    if out_split:
        C3 = torch.empty(3, M, N, dtype=torch.bfloat16, device="cuda")
        assert lib.wt_dbg_gemm_x3(P(Ad), P(Wd), P(bd), None, P(C3), M, N, K, act, P(a_pl), P(w_pl), 1, _stream()) == 0
        torch.cuda.synchronize()
        got = C3.double().sum(0).cpu()          # b1 + b2 + b3, exact in fp64
    else:
        C = R.cuda().clone() if use_res else torch.empty(M, N, device="cuda")
        assert lib.wt_dbg_gemm_x3(P(Ad), P(Wd), P(bd), P(C) if use_res else None, P(C), M, N, K, act, P(a_pl), P(w_pl), 0, _stream()) == 0
        torch.cuda.synchronize()
        got = C.cpu().double()
    # the planes really are the exact split: plane sum == the fp32 operand to 2^-24 relative of each element
    A3 = a_pl.view(3, M, K).double().sum(0).cpu()
    assert ((A3 - A.double()).abs() <= A.double().abs() * 2.0 ** -24).all()
    Cn = R.cuda().clone() if use_res else torch.empty(M, N, device="cuda")
    assert lib.wt_dbg_gemm(P(Ad), K, P(Wd), P(bd), P(Cn) if use_res else None, P(Cn), M, N, K, act, _stream()) == 0
    torch.cuda.synchronize()
    err_x3 = (got - ref).abs().max().item()
    err_native = (Cn.cpu().double() - ref).abs().max().item()
    scale = max(1.0, ref.abs().max().item())
    print(f"gemm_x3 {M}x{N}x{K}: |err| {err_x3:.3e} (native fp32 MFMA {err_native:.3e}, |ref| {scale:.1f})")
    slack = 2.0 ** -24 * scale if out_split else 0.0     # the three-plane output is itself rounded to 2^-27 .. 2^-24 of each element
    assert err_x3 <= 1.5 * err_native + slack + 1e-7 * scale, (err_x3, err_native)
    assert err_x3 < 2e-5 * scale


@pytest.mark.parametrize("M,N,K,act,use_res", [
    # more than one round of 128x128 tiles (> 768 workgroups) with 3 (= pipeline depth), 4 and 16 K-steps per tile and ragged last
    # tile row / column: interior sub-tiles take the branch-free epilogue, edge sub-tiles the generic one, in the same launch;
    # every epilogue kind wt_dbg_gemm can reach (plain, GELU, residual in place)
    (4000, 3200, 48, 0, False), (4100, 3210, 64, 1, False), (5000, 2600, 256, 0, True), (12000, 1100, 64, 0, True),
    # (one-round launches of 752 tiles with one and two K-steps per tile: fewer K-steps than pipeline stages)
    (12000, 1000, 16, 0, False), (11900, 1024, 32, 1, False),
    # >= 2048 tiles with K % 32 == 0: the two-stage 128x128x32 form (one, two, three and five K steps: fewer than, as many as and more than
    # its pipeline stages; GELU, residual in place, ragged last row tile)
    (12000, 3072, 32, 0, False), (12000, 3072, 64, 1, False), (11990, 3072, 96, 0, True), (8200, 4096, 160, 1, False),
    # the same launches with K % 32 == 16: they must take the three-stage 128x128x16 form
    (12000, 3072, 48, 0, True)])
def test_gemm_many_tiles(lib, M, N, K, act, use_res):
    A, W, b = _rand(M, K, seed=1), _rand(N, K, seed=2, scale=K ** -0.5), _rand(N, seed=3)
    R = _rand(M, N, seed=4) if use_res else None
    ref = F.linear(A.double(), W.double(), b.double())
    if act:
        ref = F.gelu(ref)
    if use_res:
        ref = ref + R.double()
    Ad, Wd, bd = A.cuda(), W.cuda(), b.cuda()
    for rep in range(2):
        C = R.cuda().clone() if use_res else torch.full((M, N), float("nan"), device="cuda")
        assert lib.wt_dbg_gemm(P(Ad), K, P(Wd), P(bd), P(C) if use_res else None, P(C), M, N, K, act, _stream()) == 0
        torch.cuda.synchronize()
        err = (C.cpu().double() - ref).abs().max().item()
        assert err < 2e-5 * max(1.0, ref.abs().max().item()), (rep, err)


def test_gemm_probe_build_stamps(lib):
    """The probe build of the LDS-DMA GEMM (tools/gemm_stamps.py) computes the same product and leaves ordered wall-clock stamps and a
    plausible placement per workgroup."""
    M, N, K = 1000, 640, 256
    A, W, b = _rand(M, K, seed=1), _rand(N, K, seed=2, scale=K ** -0.5), _rand(N, seed=3)
    Ad, Wd, bd = A.cuda(), W.cuda(), b.cuda()
    C = torch.empty(M, N, device="cuda")
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    stamps = torch.zeros(tiles, 8, dtype=torch.int64, device="cuda")
    assert lib.wt_dbg_gemm_stamps(P(Ad), K, P(Wd), P(bd), None, P(C), M, N, K, 0, P(stamps), _stream()) == 0
    torch.cuda.synchronize()
    ref = F.linear(A.double(), W.double(), b.double())
    assert (C.cpu().double() - ref).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item())
    s = stamps.cpu().numpy()
    assert ((s[:, 1] & 0xF) < 8).all()                                  # XCC_ID
    entry, first, loop_done, drained, issued = s[:, 2], s[:, 3], s[:, 4], s[:, 5], s[:, 6]
    assert (entry > 0).all() and (entry <= first).all() and (first <= loop_done).all() and (loop_done <= issued).all() and (issued <= drained).all()
    assert (drained - entry).max() < 100 * 1000                         # 100 MHz clock: every workgroup done within a millisecond


@pytest.mark.parametrize("M,N,K,act,out_half,use_res", [
    (128, 128, 64, 0, 0, False), (200, 136, 240, 1, 1, False), (1500, 384, 384, 0, 0, True), (97, 1000, 1536, 1, 1, False),
    (3000, 128, 240, 1, 1, False), (33, 51, 8, 0, 0, False), (1024, 3072, 1024, 0, 0, True),
    (130, 70, 64, 0, 0, False), (257, 129, 128, 1, 1, False), (1500, 1024, 4096, 0, 0, True), (3000, 1024, 3072, 1, 0, False)])   # LDS-DMA kernel: ragged M/N tiles, 1..64 K-tiles
def test_gemm_f16(lib, M, N, K, act, out_half, use_res):
    A, W, b = _rand(M, K, seed=1).half(), _rand(N, K, seed=2, scale=K ** -0.5).half(), _rand(N, seed=3)
    R = _rand(M, N, seed=4) if use_res else None
    ref = F.linear(A.double(), W.double(), b.double())   # fp16 inputs are exact in fp64: only accumulation differs
    if act:
        ref = F.gelu(ref)
    if use_res:
        ref = ref + R.double()
    Ad, Wd, bd = A.cuda(), W.cuda(), b.cuda()
    C = (R.cuda().clone() if use_res else torch.empty(M, N, device="cuda"))
    if out_half:
        C = torch.empty(M, N, device="cuda", dtype=torch.float16)
    assert lib.wt_dbg_gemm_f16(P(Ad), K, P(Wd), P(bd), P(C) if use_res else None, P(C), M, N, K, act, out_half, _stream()) == 0
    torch.cuda.synchronize()
    err = (C.cpu().double() - ref).abs().max().item()
    tol = (2e-3 if out_half else 2e-5) * max(1.0, ref.abs().max().item())
    assert err < tol, err


@pytest.mark.parametrize("variant", [2, 3, 4, 5])
@pytest.mark.parametrize("M,N,K,act,out_half,use_res", [
    (130, 70, 64, 0, 0, False), (257, 129, 128, 1, 1, False), (1500, 1024, 4096, 0, 0, True), (3000, 1024, 3072, 1, 0, False),
    (1024, 3072, 1024, 0, 1, False), (300, 200, 192, 0, 0, True), (256, 128, 64, 0, 0, False), (511, 383, 640, 1, 1, False),
    # more than 256 tiles of 256x128: the persistent kernel's workgroups walk 2+ tiles each (tile switch inside the K-step stream,
    # epilogue under the next tile's DMA), with 1, 2 and 3 K-steps per tile (pipeline shorter than / equal to its depth) and ragged edges
    (8192, 1152, 64, 0, 0, False), (8192, 1152, 128, 1, 1, False), (20000, 520, 192, 0, 0, True), (9000, 1030, 256, 0, 1, False),
    # more than 256 tiles of 256x256 as well (the 256x256x32 four-stage kernel: 2, 4, 6 K-steps of 32 per tile)
    (8192, 2304, 64, 1, 1, False), (20000, 1030, 128, 0, 0, True), (12000, 1540, 192, 0, 1, False)])
def test_gemm_f16_lds_dma_kernels(lib, variant, M, N, K, act, out_half, use_res):
    """The LDS-DMA fp16 GEMM kernels (128x128 two-stage; persistent 256x128x64 three-stage and 256x256x32 four-stage with counted
    vmcnt + raw barrier) forced at sizes the shape rule would not give them: ragged M / N tiles, 1..64 K-tiles, every epilogue form."""
    A, W, b = _rand(M, K, seed=1).half(), _rand(N, K, seed=2, scale=K ** -0.5).half(), _rand(N, seed=3)
    R = _rand(M, N, seed=4) if use_res else None
    ref = F.linear(A.double(), W.double(), b.double())
    if act:
        ref = F.gelu(ref)
    if use_res:
        ref = ref + R.double()
    Ad, Wd, bd = A.cuda(), W.cuda(), b.cuda()
    C = (R.cuda().clone() if use_res else torch.empty(M, N, device="cuda"))
    if out_half:
        C = torch.empty(M, N, device="cuda", dtype=torch.float16)
    for _ in range(2):
        C0 = R.cuda().clone() if use_res else C
        assert lib.wt_dbg_gemm_f16_variant(P(Ad), K, P(Wd), P(bd), P(C0) if use_res else None, P(C0), M, N, K, act, out_half, variant, _stream()) == 0
    torch.cuda.synchronize()
    err = (C0.cpu().double() - ref).abs().max().item()
    tol = (2e-3 if out_half else 2e-5) * max(1.0, ref.abs().max().item())
    assert err < tol, err


@pytest.mark.parametrize("n", [160, 192, 256])   # 192, 256: K % 64 == 0 -> the LDS-DMA kernel (swizzled unpadded tiles)
def test_gemm_f16_identity_asymmetric(lib, n):
    A = torch.eye(n).half()
    W = (torch.arange(n * n, dtype=torch.float32).reshape(n, n) % 251 / 8.0).half()
    C = torch.empty(n, n, device="cuda")
    Ad, Wd = A.cuda(), W.cuda()
    assert lib.wt_dbg_gemm_f16(P(Ad), n, P(Wd), None, None, P(C), n, n, n, 0, 0, _stream()) == 0
    torch.cuda.synchronize()
    np.testing.assert_array_equal(C.cpu().numpy(), W.float().t().numpy())


def test_gemm_identity_asymmetric(lib):
    """A = I with an asymmetric W catches a transposed C/D fragment map (cdna guide §3)."""
    n = 160
    A = torch.eye(n)
    W = torch.arange(n * n, dtype=torch.float32).reshape(n, n) / 7.0
    C = torch.empty(n, n, device="cuda")
    Ad, Wd = A.cuda(), W.cuda()
    assert lib.wt_dbg_gemm(P(Ad), n, P(Wd), None, None, P(C), n, n, n, 0, _stream()) == 0
    torch.cuda.synchronize()
    np.testing.assert_allclose(C.cpu().numpy(), W.t().numpy(), rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("half", [False, True])
def test_gemm_operand_span_over_4gib_takes_the_64bit_path(lib, half):
    """The LDS-DMA GEMMs address their operands with 32-bit byte offsets from a scalar base; the launchers must send an operand that spans
    4 GiB or more to the register-staged kernels (64-bit addresses).  A with a row stride of 512 MiB and 9 rows spans 4.5 GiB (only K
    elements per row are ever read); a wrapped offset would stay inside the allocation and read row (m mod 8) instead of row m."""
    M, N, K = 9, 128, 64
    lda = (1 << 27) if not half else (1 << 28)                    # elements: 512 MiB per row either way
    dt = torch.float16 if half else torch.float32
    big = torch.empty(M * lda, dtype=dt, device="cuda")
    rows = _rand(M, K, seed=21).to(dt)
    big.view(M, lda)[:, :K] = rows.cuda()
    W = _rand(N, K, seed=22, scale=K ** -0.5).to(dt)
    Wd, C = W.cuda(), torch.empty(M, N, device="cuda")
    if half:
        assert lib.wt_dbg_gemm_f16(P(big), lda, P(Wd), None, None, P(C), M, N, K, 0, 0, _stream()) == 0
    else:
        assert lib.wt_dbg_gemm(P(big), lda, P(Wd), None, None, P(C), M, N, K, 0, _stream()) == 0
    torch.cuda.synchronize()
    ref = rows.double() @ W.double().t()
    assert (C.cpu().double() - ref).abs().max().item() < (2e-3 if half else 2e-5)
    del big


def test_gemm_gelu_epilogue_accuracy(lib):
    """The encoder's GELU (csrc/kernels_encoder.hip: gelu_erf, a branch-free erf fitted by tools/fit_gelu.py) measured alone: A = I makes
    the product exact, so C = gelu(W^T) element for element.  Against the float64 erf form: within 6e-7 max(|x|, 1) -- one float32 ulp
    of the result -- on a grid over [-12, 12] plus the values where the polynomial's error peaks, zero, and +-large."""
    n = 256
    x = torch.linspace(-12.0, 12.0, n * n - 8, dtype=torch.float64)
    x = torch.cat([x, torch.tensor([0.0, -0.0, 4.341, -4.341, 5.5437, -5.5437, 30.0, -30.0], dtype=torch.float64)]).float()
    W = x.reshape(n, n)
    Ad, Wd = torch.eye(n).cuda(), W.cuda()
    C = torch.empty(n, n, device="cuda")
    assert lib.wt_dbg_gemm(P(Ad), n, P(Wd), None, None, P(C), n, n, n, 1, _stream()) == 0
    torch.cuda.synchronize()
    xd = W.t().double()
    want = 0.5 * xd * (1.0 + torch.erf(xd / 2.0 ** 0.5))
    err = (C.cpu().double() - want).abs() / xd.abs().clamp(min=1.0)
    assert err.max().item() < 6e-7, (err.max().item(), xd.flatten()[err.argmax()].item())
    got = C.cpu()
    assert got[W.t() == 0].abs().max().item() == 0.0 and torch.isfinite(got).all()
    assert got[W.t() == 30.0].item() == 30.0 and got[W.t() == -30.0].abs().item() < 1e-30


@pytest.mark.parametrize("rows,d", [(7, 128), (1500, 384), (33, 1024), (5, 192), (4, 768)])
def test_layernorm(lib, rows, d):
    x, w, b = _rand(rows, d, seed=5, scale=3.0) + 0.7, _rand(d, seed=6), _rand(d, seed=7)
    y = torch.empty(rows, d, device="cuda")
    xd, wd, bd = x.cuda(), w.cuda(), b.cuda()  # keep the device copies alive across the async launch
    assert lib.wt_dbg_layernorm(P(xd), P(wd), P(bd), P(y), rows, d, _stream()) == 0
    torch.cuda.synchronize()
    ref = F.layer_norm(x.double(), (d,), w.double(), b.double(), 1e-5)
    assert (y.cpu().double() - ref).abs().max().item() < 2e-5


@pytest.mark.parametrize("B,S,H,scale", [(1, 64, 1, 1.0), (2, 96, 2, 1.0), (1, 1500, 2, 1.0), (2, 160, 3, 4.0), (1, 129, 1, 1.0),
                                         (1, 200, 1, 30.0)])
def test_encoder_attention(lib, B, S, H, scale):
    d = 64 * H
    qkv = _rand(B * S, 3 * d, seed=8, scale=scale)
    if scale >= 30.0:  # force a late running-max jump: one key dominates one query in the last KV tile
        qkv[5, :64] = 6.0
        qkv[S - 3, d:d + 64] = 6.0
    ctx = torch.empty(B * S, d, device="cuda")
    qkv_d = qkv.cuda()
    assert lib.wt_dbg_encoder_attention(P(qkv_d), P(ctx), B, S, H, _stream()) == 0
    torch.cuda.synchronize()
    t = qkv.double().view(B, S, 3, H, 64)
    q, k, v = (t[:, :, i].transpose(1, 2) for i in range(3))
    ref = (torch.softmax((q * 0.125) @ k.transpose(-1, -2), -1) @ v).transpose(1, 2).reshape(B * S, d)
    got = ctx.cpu().double()
    assert torch.isfinite(got).all()
    assert (got - ref).abs().max().item() < 3e-5 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("B,S,H,scale", [(1, 64, 1, 1.0), (2, 96, 2, 1.0), (1, 1500, 2, 1.0), (2, 160, 3, 4.0), (1, 129, 1, 1.0),
                                         (1, 200, 1, 30.0)])
def test_encoder_attention_x3_is_an_fp32_attention(lib, B, S, H, scale):
    """enc_attn_x3_kernel: both products of the attention formed on the bf16 matrix cores from exactly split operands (q, k, v planes in,
    probabilities split in registers), fp32 scores / softmax / accumulators.  The same cases and the same bar as the fp32-MFMA kernel
    (3e-5 of the output range, incl. a late running-max jump at scores in the thousands), and its error within 2x of that kernel's."""
    d = 64 * H
    qkv = _rand(B * S, 3 * d, seed=8, scale=scale)
    if scale >= 30.0:
        qkv[5, :64] = 6.0
        qkv[S - 3, d:d + 64] = 6.0
    qkv_d = qkv.cuda()
    qkv_pl = torch.empty(3 * B * S * 3 * d, dtype=torch.bfloat16, device="cuda")
    ctx_pl = torch.empty(3, B * S, d, dtype=torch.bfloat16, device="cuda")
    ctx = torch.empty(B * S, d, device="cuda")
    assert lib.wt_dbg_encoder_attention_x3(P(qkv_d), P(qkv_pl), P(ctx_pl), B, S, H, 0, _stream()) == 0
    assert lib.wt_dbg_encoder_attention(P(qkv_d), P(ctx), B, S, H, _stream()) == 0
    torch.cuda.synchronize()
    t = qkv.double().view(B, S, 3, H, 64)
    q, k, v = (t[:, :, i].transpose(1, 2) for i in range(3))
    ref = (torch.softmax((q * 0.125) @ k.transpose(-1, -2), -1) @ v).transpose(1, 2).reshape(B * S, d)
    got = ctx_pl.double().sum(0).cpu()
    assert torch.isfinite(got).all()
    err, err_native = (got - ref).abs().max().item(), (ctx.cpu().double() - ref).abs().max().item()
    print(f"enc_attn_x3 B={B} S={S} H={H} scale={scale}: |err| {err:.2e} (fp32-MFMA kernel {err_native:.2e}, range {ref.abs().max().item():.1f})")
    bar = 3e-5 * max(1.0, ref.abs().max().item())
    assert err < bar
    assert err <= 2.0 * err_native + 0.1 * bar


def test_encoder_attention_three_plane_output_equals_the_fp32_output(lib):
    """The x3 encoder path takes the attention context as three bf16 planes (the out-projection's A operand): b1 + b2 + b3 must equal the
    fp32 context of the plain kernel to 2^-24 of each element (the same arithmetic, only the stores differ)."""
    B, S, H = 2, 333, 3
    d = 64 * H
    qkv = _rand(B * S, 3 * d, seed=8).cuda()
    ctx = torch.empty(B * S, d, device="cuda")
    planes = torch.empty(3, B * S, d, dtype=torch.bfloat16, device="cuda")
    assert lib.wt_dbg_encoder_attention(P(qkv), P(ctx), B, S, H, _stream()) == 0
    assert lib.wt_dbg_encoder_attention_split(P(qkv), P(planes), B, S, H, _stream()) == 0
    torch.cuda.synchronize()
    got, want = planes.double().sum(0), ctx.double()
    assert ((got - want).abs() <= want.abs() * 2.0 ** -24).all()


@pytest.mark.parametrize("B,S,H,scale", [(1, 64, 1, 1.0), (2, 96, 2, 1.0), (1, 1500, 2, 1.0), (2, 160, 3, 3.0), (1, 129, 1, 1.0), (1, 200, 1, 12.0)])
def test_encoder_attention_f16(lib, B, S, H, scale):
    d = 64 * H
    qkv = (_rand(B * S, 3 * d, seed=8, scale=scale)).half()
    if scale >= 12.0:  # late running-max jump in the last KV tile
        qkv[5, :64] = 3.0
        qkv[S - 3, d:d + 64] = 3.0
    ctx = torch.empty(B * S, d, device="cuda", dtype=torch.float16)
    qkv_d = qkv.cuda()
    assert lib.wt_dbg_encoder_attention_f16(P(qkv_d), P(ctx), B, S, H, _stream()) == 0
    torch.cuda.synchronize()
    t = qkv.double().view(B, S, 3, H, 64)            # the fp16 inputs are exact in fp64
    q, k, v = (t[:, :, i].transpose(1, 2) for i in range(3))
    ref = (torch.softmax((q * 0.125) @ k.transpose(-1, -2), -1) @ v).transpose(1, 2).reshape(B * S, d)
    got = ctx.cpu().double()
    assert torch.isfinite(got).all()
    # P is rounded to fp16 before the second product and the output is fp16: ~1e-3 relative
    assert (got - ref).abs().max().item() < 4e-3 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("B", [1, 2, 3, 4, 5, 8, 9, 12, 16])
@pytest.mark.parametrize("N,K,xmode,act,use_res", [
    (384, 384, 1, 0, False), (1152, 384, 1, 0, False), (1536, 384, 1, 1, False), (384, 1536, 0, 0, True),
    (1024, 1024, 1, 0, True), (1024, 4096, 0, 0, True), (768, 3072, 0, 1, False), (1001, 128, 1, 0, False),
    (130, 2048, 0, 0, False), (7, 512, 0, 0, False),
    (2304, 768, 1, 0, False), (768, 768, 0, 0, True), (1536, 512, 1, 1, False), (512, 2048, 0, 0, True)])   # small.en / base.en widths
def test_skinny(lib, B, N, K, xmode, act, use_res):
    X, W, b = _rand(B, K, seed=9, scale=2.0) + 0.3, _rand(N, K, seed=10, scale=K ** -0.5), _rand(N, seed=11)
    g, be = _rand(K, seed=12) + 1.0, _rand(K, seed=13)
    R = _rand(B, N, seed=14) if use_res else None
    xin = F.layer_norm(X.double(), (K,), g.double(), be.double(), 1e-5) if xmode == 1 else X.double()
    ref = (F.linear(xin, W.double(), b.double())) * 0.5
    if act:
        ref = F.gelu(ref)
    if use_res:
        ref = ref + R.double()
    Y = R.cuda().clone() if use_res else torch.full((B, N), float("nan"), device="cuda")
    Xd, gd, bed, Wd, bd = X.cuda(), g.cuda(), be.cuda(), W.cuda(), b.cuda()
    rc = lib.wt_dbg_skinny(P(Xd), P(gd), P(bed), P(Wd), P(bd), P(Y) if use_res else None, P(Y),
                           B, N, K, xmode, act, 0.5, _stream())
    assert rc == 0
    torch.cuda.synchronize()
    err = (Y.cpu().double() - ref).abs().max().item()
    assert err < 3e-5 * max(1.0, ref.abs().max().item()), err


@pytest.mark.parametrize("B,H,cap,length,n_split", [(1, 2, 40, 1, 1), (3, 2, 40, 17, 2), (2, 6, 1500, 1500, 8), (8, 2, 448, 447, 3),
                                                    (1, 1, 96, 96, 16), (2, 3, 160, 5, 4), (1, 2, 64, 63, 1),
                                                    # around the eight-wave kernel's stream-count switch (128 keys per split): 16 streams below,
                                                    # 32 from there on; fewer keys than streams; one key per stream
                                                    (2, 2, 448, 127, 1), (2, 2, 448, 128, 1), (2, 2, 448, 129, 1), (3, 2, 448, 255, 2),
                                                    (3, 2, 448, 256, 2), (3, 2, 448, 257, 2), (1, 4, 448, 31, 1), (1, 4, 448, 32, 1), (2, 2, 448, 33, 2)])
def test_decode_attention(lib, B, H, cap, length, n_split):
    d = 64 * H
    q = _rand(B, d, seed=15) * 0.5
    k, v = _rand(B, H, cap, 64, seed=16), _rand(B, H, cap, 64, seed=17)
    part = torch.full((B, H, n_split, 68), float("nan"), device="cuda")
    cnt = torch.zeros(B, H, dtype=torch.int32, device="cuda")
    out = torch.full((B, d), float("nan"), device="cuda")
    qd, kd, vd = q.cuda(), k.cuda(), v.cuda()
    for _ in range(3):  # the arrival tickets must re-arm themselves between launches
        assert lib.wt_dbg_decode_attention(P(qd), P(kd), P(vd), P(part), P(cnt), P(out), B, H, cap, length, n_split, _stream()) == 0
    torch.cuda.synchronize()
    assert int(cnt.abs().sum()) == 0
    qh = q.double().view(B, H, 1, 64)
    att = torch.softmax(qh @ k.double()[:, :, :length].transpose(-1, -2), -1)
    ref = (att @ v.double()[:, :, :length]).reshape(B, d)
    got = out.cpu().double()
    assert torch.isfinite(got).all()
    assert (got - ref).abs().max().item() < 2e-5


@pytest.mark.parametrize("B,H,cap,length,n_split", [(1, 2, 40, 17, 1), (3, 6, 1500, 1500, 2), (8, 16, 1500, 1500, 2), (2, 3, 160, 5, 4),
                                                    (5, 8, 96, 96, 3)])
def test_decode_attention_folded_query(lib, B, H, cap, length, n_split):
    """Cross-attention with the folded query (builder.py:_fold_cross_query): the kernel finishes
    q = (u - mean(h1).r).rstd(h1) + t from the LayerNorm statistics of the residual row; d = 64 H covers 128..1024."""
    d = 64 * H
    u, h1 = _rand(B, d, seed=21) * 0.5, _rand(B, d, seed=22) * 2.0 + 0.7
    r, t = _rand(d, seed=23), _rand(d, seed=24) * 0.3
    k, v = _rand(B, H, cap, 64, seed=16), _rand(B, H, cap, 64, seed=17)
    part = torch.full((B, H, n_split, 68), float("nan"), device="cuda")
    cnt = torch.zeros(B, H, dtype=torch.int32, device="cuda")
    out = torch.full((B, d), float("nan"), device="cuda")
    ud, hd, rd, td, kd, vd = u.cuda(), h1.cuda(), r.cuda(), t.cuda(), k.cuda(), v.cuda()
    for _ in range(2):
        assert lib.wt_dbg_decode_attention_folded(P(ud), P(kd), P(vd), P(part), P(cnt), P(out), P(hd), P(rd), P(td), B, H, cap, length,
                                                  n_split, _stream()) == 0
    torch.cuda.synchronize()
    h64 = h1.double()
    mu = h64.mean(1, keepdim=True)
    rstd = (h64.var(1, unbiased=False, keepdim=True) + 1e-5).rsqrt()
    q = ((u.double() - mu * r.double()) * rstd + t.double()).view(B, H, 1, 64)
    att = torch.softmax(q @ k.double()[:, :, :length].transpose(-1, -2), -1)
    ref = (att @ v.double()[:, :, :length]).reshape(B, d)
    got = out.cpu().double()
    assert torch.isfinite(got).all()
    assert (got - ref).abs().max().item() < 3e-5


@pytest.mark.parametrize("B", [1, 3, 4, 8, 11, 16])
@pytest.mark.parametrize("d", [128, 384, 512, 768, 1024])
def test_skinny_pair(lib, B, d):
    """One launch = out-projection + residual (K = d, LDS-staged activations) and the concatenated-input GEMM
    y = W.[a ; h] + c (K = 2d, two activation buffers): both halves against fp64 torch."""
    a, h = _rand(B, d, seed=31), _rand(B, d, seed=32)
    Wo, bo = _rand(d, d, seed=33, scale=d ** -0.5), _rand(d, seed=34)
    Wf, c = _rand(d, 2 * d, seed=35, scale=(2 * d) ** -0.5), _rand(d, seed=36)
    ad, hd, Wod, bod, Wfd, cd = a.cuda(), h.cuda(), Wo.cuda(), bo.cuda(), Wf.cuda(), c.cuda()
    h1 = torch.full((B, d), float("nan"), device="cuda")
    uo = torch.full((B, d), float("nan"), device="cuda")
    assert lib.wt_dbg_skinny_pair(P(ad), P(Wod), P(bod), P(hd), P(h1), d, d, P(ad), P(hd), P(Wfd), P(cd), P(uo), d, 2 * d, B, _stream()) == 0
    torch.cuda.synchronize()
    ref1 = h.double() + a.double() @ Wo.double().T + bo.double()
    ref2 = torch.cat([a, h], 1).double() @ Wf.double().T + c.double()
    assert (h1.cpu().double() - ref1).abs().max().item() < 2e-5
    assert (uo.cpu().double() - ref2).abs().max().item() < 2e-5
    assert torch.equal(hd.cpu(), h)   # the residual input is read-only (h1 goes to the other buffer)


@pytest.mark.parametrize("B,H,cap,length,n_split", [(8, 16, 1500, 1500, 2), (3, 6, 1500, 1500, 2), (1, 8, 1500, 1500, 2), (2, 12, 200, 199, 2),
                                                    (16, 16, 1500, 1500, 2), (5, 2, 96, 96, 2), (11, 16, 448, 3, 2)])
def test_attention_with_deferred_merge(lib, B, H, cap, length, n_split):
    """The decode step's cross-attention -> out-projection pair: the attention kernel leaves its split partials, the GEMV merges
    them while staging.  Against fp64 torch, and bitwise against the attention kernel's own (ticket) merge + the plain GEMV."""
    d = 64 * H
    q = _rand(B, d, seed=41) * 0.5
    k, v = _rand(B, H, cap, 64, seed=42), _rand(B, H, cap, 64, seed=43)
    W, bias, resid = _rand(d, d, seed=44, scale=d ** -0.5), _rand(d, seed=45), _rand(B, d, seed=46)
    qd, kd, vd, Wd, bd, rd = q.cuda(), k.cuda(), v.cuda(), W.cuda(), bias.cuda(), resid.cuda()
    part = torch.full((B, H, n_split, 68), float("nan"), device="cuda")
    y = torch.full((B, d), float("nan"), device="cuda")
    assert lib.wt_dbg_attention_then_projection(P(qd), P(kd), P(vd), P(part), P(Wd), P(bd), P(rd), P(y), B, H, cap, length, n_split, _stream()) == 0
    # the two-kernel reference path of the same library: attention with its own merge, then the plain GEMV
    cnt = torch.zeros(B, H, dtype=torch.int32, device="cuda")
    att = torch.full((B, d), float("nan"), device="cuda")
    y2 = torch.full((B, d), float("nan"), device="cuda")
    part2 = torch.full((B, H, n_split, 68), float("nan"), device="cuda")
    assert lib.wt_dbg_decode_attention(P(qd), P(kd), P(vd), P(part2), P(cnt), P(att), B, H, cap, length, n_split, _stream()) == 0
    assert lib.wt_dbg_skinny(P(att), None, None, P(Wd), P(bd), P(rd), P(y2), B, d, d, 4, 0, 1.0, _stream()) == 0
    torch.cuda.synchronize()
    assert torch.equal(y.cpu(), y2.cpu())
    qh = q.double().view(B, H, 1, 64)
    a64 = (torch.softmax(qh @ k.double()[:, :, :length].transpose(-1, -2), -1) @ v.double()[:, :, :length]).reshape(B, d)
    ref = resid.double() + a64 @ W.double().T + bias.double()
    assert (y.cpu().double() - ref).abs().max().item() < 3e-5


@pytest.mark.parametrize("B,H,mean,outlier,mixed_gamma", [
    (8, 16, 0.0, 0.0, False), (8, 16, 3.0, 100.0, False), (8, 16, 10.0, 100.0, True), (3, 6, 10.0, 100.0, True),
    (8, 12, 5.0, 300.0, True), (16, 16, 10.0, 100.0, True), (8, 16, 20.0, 0.0, False)])   # last: mean = 20 sigma, no outlier (worst case for the fold)
def test_folded_query_adversarial_rows(lib, B, H, mean, outlier, mixed_gamma):
    """The folded cross-attention query END TO END (builder._fold_cross_query -> skinny_pair_kernel -> dec_attn_kernel) on the
    rows real Whisper decoders have and N(0, s^2) tests do not: residual rows whose mean is `mean` x their standard deviation,
    two massive channels at `outlier` x the rest, LayerNorm gammas of mixed sign and magnitude.  Reference: the reference's own
    order of operations in fp64 -- h1 = h + Wo.a + bo, q = s.(Wq.LN(h1) + bq) (HF modeling_whisper.py:472, 727-735) -- then
    softmax(q K^T) V.  q = (u - mean.r).rstd + t subtracts two terms of size ~|mean|.|r|; measured (fp32 emulation, d = 1024):
    error of q 5e-7 at mean 0, 2e-6 at mean 10 sigma, 1e-5 at mean 50 sigma -- tolerance here 1e-5 on q-equivalent terms,
    which keeps logits far inside the 1e-3 budget."""
    import whisper_trtllm_amd as w
    d, S = 64 * H, 1500
    g = torch.Generator().manual_seed(1234 + B + H)
    rn = lambda *s: torch.randn(*s, generator=g)
    wq, wo = rn(d, d) * 1.6 / d ** 0.5, rn(d, d) * 1.6 / d ** 0.5
    bq, bo, beta = rn(d) * 0.1, rn(d) * 0.1, rn(d) * 0.1
    gamma = 1.0 + 0.1 * rn(d)
    if mixed_gamma:
        gamma = gamma * torch.tensor([-3.0, -1.0, 0.05, 1.0, 4.0])[torch.randint(0, 5, (d,), generator=g)]
    a, h = rn(B, d), rn(B, d) + mean
    if outlier:
        h[:, 7] *= outlier
        h[:, d // 3] = -3.0 * outlier
    k, v = rn(B, H, S, 64), rn(B, H, S, 64)
    Wf, c, r, t = (torch.from_numpy(x) for x in w.builder.Builder._fold_cross_query(
        wq.numpy(), bq.numpy(), gamma.numpy(), beta.numpy(), wo.numpy(), bo.numpy()))
    ad, hd, Wod, bod, Wfd, cd, rd, td, kd, vd = (x.cuda() for x in (a, h, wo, bo, Wf, c, r, t, k, v))
    h1 = torch.full((B, d), float("nan"), device="cuda")
    u = torch.full((B, d), float("nan"), device="cuda")
    n_split = 2
    part = torch.full((B, H, n_split, 68), float("nan"), device="cuda")
    cnt = torch.zeros(B, H, dtype=torch.int32, device="cuda")
    out = torch.full((B, d), float("nan"), device="cuda")
    assert lib.wt_dbg_skinny_pair(P(ad), P(Wod), P(bod), P(hd), P(h1), d, d, P(ad), P(hd), P(Wfd), P(cd), P(u), d, 2 * d, B, _stream()) == 0
    assert lib.wt_dbg_decode_attention_folded(P(u), P(kd), P(vd), P(part), P(cnt), P(out), P(h1), P(rd), P(td), B, H, S, S, n_split,
                                              _stream()) == 0
    torch.cuda.synchronize()
    h64 = h.double() + a.double() @ wo.double().T + bo.double()
    ln = F.layer_norm(h64, (d,), gamma.double(), beta.double(), 1e-5)
    q64 = 0.125 * (ln @ wq.double().T + bq.double())
    ref = (torch.softmax(q64.view(B, H, 1, 64) @ k.double().transpose(-1, -2), -1) @ v.double()).reshape(B, d)
    assert (h1.cpu().double() - h64).abs().max().item() < 3e-5 * max(1.0, h64.abs().max().item())
    # the query itself, reconstructed from the kernel's inputs exactly as the kernel finishes it (fp64 statistics of ITS h1)
    mu, var = h1.cpu().double().mean(1, keepdim=True), h1.cpu().double().var(1, unbiased=False, keepdim=True)
    q_fold = (u.cpu().double() - mu * r.double()) * (var + 1e-5).rsqrt() + t.double()
    q_err = (q_fold - q64).abs().max().item()
    assert q_err < 1e-5, f"folded query differs from s.(Wq.LN(h1)+bq) by {q_err:.2e}"
    got = out.cpu().double()
    assert torch.isfinite(got).all()
    assert (got - ref).abs().max().item() < 5e-5


@pytest.mark.parametrize("B,H,cap,length", [(8, 16, 448, 447), (8, 16, 448, 1), (8, 16, 448, 2), (3, 8, 448, 100), (1, 16, 64, 33), (8, 6, 448, 200),
                                            (16, 16, 448, 447), (5, 12, 448, 17)])
def test_self_attention_two_splits_merged_by_both_halves_of_the_pair(lib, B, H, cap, length):
    """Decode self-attention over two key splits, merge deferred into the pair launch: the out-projection half merges the partials while
    it stages its rows, the folded-query half stages the merged `a` for its first K/2 columns and reads h directly for the rest.
    Against fp64 torch; includes length 1 (the second split has no key) and every batch-width instantiation."""
    d = 64 * H
    q = _rand(B, d, seed=51) * 0.5
    k, v = _rand(B, H, cap, 64, seed=52), _rand(B, H, cap, 64, seed=53)
    Wo, bo = _rand(d, d, seed=54, scale=d ** -0.5), _rand(d, seed=55)
    Wf, c = _rand(d, 2 * d, seed=56, scale=(2 * d) ** -0.5), _rand(d, seed=57)
    h = _rand(B, d, seed=58)
    qd, kd, vd, Wod, bod, Wfd, cd, hd = (t.cuda() for t in (q, k, v, Wo, bo, Wf, c, h))
    part = torch.full((B, H, 2, 68), float("nan"), device="cuda")
    h1 = torch.full((B, d), float("nan"), device="cuda")
    u = torch.full((B, d), float("nan"), device="cuda")
    assert lib.wt_dbg_self_attention_then_pair(P(qd), P(kd), P(vd), P(part), P(Wod), P(bod), P(hd), P(h1), P(Wfd), P(cd), P(u), B, H, cap, length,
                                               _stream()) == 0
    torch.cuda.synchronize()
    qh = q.double().view(B, H, 1, 64)
    a = (torch.softmax(qh @ k.double()[:, :, :length].transpose(-1, -2), -1) @ v.double()[:, :, :length]).reshape(B, d)
    ref1 = h.double() + a @ Wo.double().T + bo.double()
    ref2 = torch.cat([a, h.double()], 1) @ Wf.double().T + c.double()
    assert torch.isfinite(h1).all() and torch.isfinite(u).all()
    assert (h1.cpu().double() - ref1).abs().max().item() < 3e-5
    assert (u.cpu().double() - ref2).abs().max().item() < 3e-5


# ----------------------------------------------------------------------------- fp16 decoder engines: half weights / half K/V caches
# The half operands are exact in fp64, so the references below differ from the kernels only by fp32 accumulation order: the same
# tolerances as the fp32 kernels.
@pytest.mark.parametrize("B", [1, 2, 3, 4, 5, 8, 9, 12, 16])
@pytest.mark.parametrize("N,K,xmode,act,use_res", [
    (384, 384, 1, 0, False), (1152, 384, 1, 0, False), (1536, 384, 1, 1, False), (384, 1536, 0, 0, True),
    (1024, 1024, 1, 0, True), (1024, 4096, 0, 0, True), (768, 3072, 0, 1, False), (1001, 128, 1, 0, False),
    (130, 2048, 0, 0, False), (7, 512, 0, 0, False), (51864, 384, 1, 0, False),
    (2304, 768, 1, 0, False), (768, 768, 0, 0, True), (1536, 512, 1, 1, False), (512, 2048, 0, 0, True), (320, 192, 1, 1, False), (192, 320, 0, 0, True)])
def test_skinny_half_weights(lib, B, N, K, xmode, act, use_res):
    X, W, b = _rand(B, K, seed=9, scale=2.0) + 0.3, _rand(N, K, seed=10, scale=K ** -0.5).half(), _rand(N, seed=11)
    g, be = _rand(K, seed=12) + 1.0, _rand(K, seed=13)
    R = _rand(B, N, seed=14) if use_res else None
    xin = F.layer_norm(X.double(), (K,), g.double(), be.double(), 1e-5) if xmode == 1 else X.double()
    ref = (F.linear(xin, W.double(), b.double())) * 0.5
    if act:
        ref = F.gelu(ref)
    if use_res:
        ref = ref + R.double()
    Xd, gd, bed, Wd, bd = X.cuda(), g.cuda(), be.cuda(), W.cuda(), b.cuda()
    for nt in (0, 4):   # default-policy and non-temporal weight loads
        Y = R.cuda().clone() if use_res else torch.full((B, N), float("nan"), device="cuda")
        rc = lib.wt_dbg_skinny_f16(P(Xd), P(gd), P(bed), P(Wd), P(bd), P(Y) if use_res else None, P(Y), B, N, K, xmode | nt, act, 0.5, _stream())
        assert rc == 0
        torch.cuda.synchronize()
        err = (Y.cpu().double() - ref).abs().max().item()
        assert err < 3e-5 * max(1.0, ref.abs().max().item()), (nt, err)


@pytest.mark.parametrize("folded", [False, True])
@pytest.mark.parametrize("B,H,cap,length,n_split", [(1, 2, 40, 1, 1), (3, 2, 40, 17, 2), (2, 6, 1500, 1500, 8), (8, 2, 448, 447, 3), (1, 1, 96, 96, 16),
                                                    (2, 3, 160, 5, 4), (1, 2, 64, 63, 1), (8, 16, 1500, 1500, 2), (16, 16, 1500, 1500, 1), (5, 8, 96, 33, 3)])
def test_decode_attention_half_caches(lib, B, H, cap, length, n_split, folded):
    """32 online-softmax streams of 8 lanes x 8 head dims over IEEE-half K/V caches, fp32 scores / softmax / accumulators."""
    d = 64 * H
    q = _rand(B, d, seed=15) * 0.5
    h1 = _rand(B, d, seed=22) * 2.0 + 0.7
    r, t = _rand(d, seed=23), _rand(d, seed=24) * 0.3
    k, v = _rand(B, H, cap, 64, seed=16).half(), _rand(B, H, cap, 64, seed=17).half()
    part = torch.full((B, H, n_split, 68), float("nan"), device="cuda")
    cnt = torch.zeros(B, H, dtype=torch.int32, device="cuda")
    out = torch.full((B, d), float("nan"), device="cuda")
    qd, kd, vd, hd, rd, td = q.cuda(), k.cuda(), v.cuda(), h1.cuda(), r.cuda(), t.cuda()
    for _ in range(3):  # the arrival tickets must re-arm themselves between launches
        assert lib.wt_dbg_decode_attention_f16(P(qd), P(kd), P(vd), P(part), P(cnt), P(out), P(hd) if folded else None, P(rd) if folded else None,
                                               P(td) if folded else None, B, H, cap, length, n_split, _stream()) == 0
    torch.cuda.synchronize()
    assert int(cnt.abs().sum()) == 0
    q64 = q.double()
    if folded:
        h64 = h1.double()
        mu = h64.mean(1, keepdim=True)
        rstd = (h64.var(1, unbiased=False, keepdim=True) + 1e-5).rsqrt()
        q64 = (q64 - mu * r.double()) * rstd + t.double()
    att = torch.softmax(q64.view(B, H, 1, 64) @ k.double()[:, :, :length].transpose(-1, -2), -1)
    ref = (att @ v.double()[:, :, :length]).reshape(B, d)
    got = out.cpu().double()
    assert torch.isfinite(got).all()
    assert (got - ref).abs().max().item() < 3e-5


@pytest.mark.parametrize("B,H,cap,length", [(8, 16, 1500, 1500), (3, 6, 1500, 1500), (2, 12, 200, 199), (16, 16, 1500, 1500), (5, 2, 96, 96), (11, 16, 448, 3)])
def test_attention_with_deferred_merge_half(lib, B, H, cap, length):
    """fp16 engine's cross-attention -> out-projection pair: half caches, two key splits left for the half-weight GEMV to merge."""
    d = 64 * H
    q = _rand(B, d, seed=41) * 0.5
    k, v = _rand(B, H, cap, 64, seed=42).half(), _rand(B, H, cap, 64, seed=43).half()
    W, bias, resid = _rand(d, d, seed=44, scale=d ** -0.5).half(), _rand(d, seed=45), _rand(B, d, seed=46)
    qd, kd, vd, Wd, bd, rd = q.cuda(), k.cuda(), v.cuda(), W.cuda(), bias.cuda(), resid.cuda()
    part = torch.full((B, H, 2, 68), float("nan"), device="cuda")
    y = torch.full((B, d), float("nan"), device="cuda")
    assert lib.wt_dbg_attention_then_projection_f16(P(qd), P(kd), P(vd), P(part), P(Wd), P(bd), P(rd), P(y), B, H, cap, length, _stream()) == 0
    torch.cuda.synchronize()
    qh = q.double().view(B, H, 1, 64)
    a64 = (torch.softmax(qh @ k.double()[:, :, :length].transpose(-1, -2), -1) @ v.double()[:, :, :length]).reshape(B, d)
    ref = resid.double() + a64 @ W.double().T + bias.double()
    assert (y.cpu().double() - ref).abs().max().item() < 3e-5


@pytest.mark.parametrize("B", [1, 3, 4, 8, 11, 16])
@pytest.mark.parametrize("d", [128, 384, 512, 768, 1024])
def test_skinny_pair_half_weights(lib, B, d):
    a, h = _rand(B, d, seed=31), _rand(B, d, seed=32)
    Wo, bo = _rand(d, d, seed=33, scale=d ** -0.5).half(), _rand(d, seed=34)
    Wf, c = _rand(d, 2 * d, seed=35, scale=(2 * d) ** -0.5).half(), _rand(d, seed=36)
    ad, hd, Wod, bod, Wfd, cd = a.cuda(), h.cuda(), Wo.cuda(), bo.cuda(), Wf.cuda(), c.cuda()
    h1 = torch.full((B, d), float("nan"), device="cuda")
    uo = torch.full((B, d), float("nan"), device="cuda")
    assert lib.wt_dbg_skinny_pair_f16(P(ad), P(Wod), P(bod), P(hd), P(h1), d, d, P(ad), P(hd), P(Wfd), P(cd), P(uo), d, 2 * d, B, _stream()) == 0
    torch.cuda.synchronize()
    ref1 = h.double() + a.double() @ Wo.double().T + bo.double()
    ref2 = torch.cat([a, h], 1).double() @ Wf.double().T + c.double()
    assert (h1.cpu().double() - ref1).abs().max().item() < 2e-5
    assert (uo.cpu().double() - ref2).abs().max().item() < 2e-5


@pytest.mark.parametrize("out_half", [1, 0])
@pytest.mark.parametrize("B,H,rows_total,rows,cap,seq_off", [(2, 2, 96, 96, 96, 0), (3, 6, 1500, 1500, 1500, 0), (1, 2, 96, 85, 96, 11), (2, 3, 160, 160, 160, 0),
                                                             (1, 16, 1500, 1, 1500, 1499), (8, 16, 1500, 1500, 1500, 0)])
def test_cross_kv_projection_f16(lib, B, H, rows_total, rows, cap, seq_off, out_half):
    """The fp16 GEMM writing head-split K / V caches [b][h][cap][64] (fp16 resident caches, or the f32 caches of the Session path);
    partial row ranges = the engine-only 'partial cross cache' protocol (model.py:264-272).  Untouched cache rows keep their content."""
    d = 64 * H
    A = _rand(B, rows_total, d, seed=51).half()
    W, bias = _rand(2 * d, d, seed=52, scale=d ** -0.5).half(), _rand(2 * d, seed=53)
    dt = torch.float16 if out_half else torch.float32
    kc = torch.full((B, H, cap, 64), 7.0, device="cuda", dtype=dt)
    vc = torch.full((B, H, cap, 64), 7.0, device="cuda", dtype=dt)
    Ad, Wd, bd = A.cuda(), W.cuda(), bias.cuda()
    assert lib.wt_dbg_gemm_f16_kv(P(Ad), rows_total, P(Wd), P(bd), P(kc), P(vc), B, rows, H, cap, seq_off, out_half, _stream()) == 0
    torch.cuda.synchronize()
    y = F.linear(A[:, :rows].double(), W.double(), bias.double())                       # [B, rows, 2d]
    kr = y[..., :d].view(B, rows, H, 64).transpose(1, 2)
    vr = y[..., d:].view(B, rows, H, 64).transpose(1, 2)
    tol = (2e-3 if out_half else 2e-5) * max(1.0, float(y.abs().max()))
    for got, ref in ((kc, kr), (vc, vr)):
        g = got.cpu().double()
        assert (g[:, :, seq_off:seq_off + rows] - ref).abs().max().item() < tol
        keep = torch.ones(cap, dtype=torch.bool)
        keep[seq_off:seq_off + rows] = False
        assert (g[:, :, keep] == 7.0).all()
