"""GPU: the XCD-fused decoder-layer kernels (csrc/kernels_decoder_xcd.hip) against fp64 torch, kernel by kernel.

Every case is launched several times with CHANGING inputs at the same addresses and a step bump in between (the flag epoch),
so a consumer that read a stale L1 / L2 line of the previous launch would reproduce the previous result and fail."""
import ctypes

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

XCD_SYNC_WORDS = 3 * 8 * 32


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


def test_census_blocks_b_and_b_plus_8_share_an_xcd(lib):
    """The placement the fused kernels are FAST under (never what makes them correct: every barrier re-checks HW_REG_XCC_ID)."""
    out = torch.full((256,), -1, dtype=torch.int32, device="cuda")
    for _ in range(3):
        assert lib.wt_dbg_xcd_census(P(out), 256, _stream()) == 0
        torch.cuda.synchronize()
        x = out.cpu().view(32, 8)
        assert (x >= 0).all() and (x < 8).all()
        assert (x == x[0:1]).all(), "workgroups b and b+8 did not land on the same XCD"
        assert sorted(x[0].tolist()) == list(range(8)), "a 256-workgroup launch did not cover the 8 XCDs evenly"


@pytest.mark.parametrize("B", [1, 3, 8])
@pytest.mark.parametrize("d,Fd", [(1024, 4096), (512, 2048), (512, 1024), (1024, 1536)])
@pytest.mark.parametrize("n_parts", [0, 8])
def test_xcd_ffn(lib, B, d, Fd, n_parts):
    g = torch.Generator().manual_seed(7 + B + d + Fd + n_parts)
    rn = lambda *s: torch.randn(*s, generator=g)
    W1, b1, W2 = rn(Fd, d) * d ** -0.5, rn(Fd) * 0.1, rn(d, Fd) * Fd ** -0.5
    lw, lb, pb = 1.0 + 0.1 * rn(d), 0.1 * rn(d), 0.1 * rn(d)
    W1d, b1d, W2d, lwd, lbd, pbd = (t.cuda() for t in (W1, b1, W2, lw, lb, pb))
    st = torch.zeros(8, dtype=torch.int32, device="cuda")
    sync = torch.zeros(XCD_SYNC_WORDS, dtype=torch.int32, device="cuda")
    hx = torch.full((8, B, d), float("nan"), device="cuda")
    fx = torch.full((8, B, Fd // 8), float("nan"), device="cuda")
    parts_out = torch.full((8, B, d), float("nan"), device="cuda")
    hb = torch.empty(B, d, device="cuda")
    pin = torch.empty(8, B, d, device="cuda")
    for it in range(6):
        hbase, parts = rn(B, d) * 2.0 + 0.3, rn(8, B, d)
        hb.copy_(hbase)
        pin.copy_(parts)
        if n_parts and it >= 3:   # in-place form of the decode step: the base is the group copies left by the previous launch
            hx.copy_(hbase.unsqueeze(0).expand(8, B, d))
            base, stride = hx, B * d
        else:
            base, stride = hb, 0
        rc = lib.wt_dbg_xcd_ffn(P(base), stride, P(pbd), P(pin) if n_parts else None, n_parts, P(hx), P(lwd), P(lbd), P(W1d), P(b1d), P(W2d),
                                P(fx), P(parts_out), P(st), P(sync), None, B, d, Fd, _stream())
        assert rc == 0
        assert lib.wt_dbg_bump_step(P(st), _stream()) == 0
        torch.cuda.synchronize()
        assert int(st[6]) == 0, f"xcd_err = {int(st[6])} (1 = timeout, 2 = placement)"
        h = hbase.double() + pb.double() + (parts.double().sum(0) if n_parts else 0.0)
        for grp in range(8):
            assert (hx[grp].cpu().double() - h).abs().max().item() < 1e-5
        f = F.gelu(F.layer_norm(h, (d,), lw.double(), lb.double(), 1e-5) @ W1.double().T + b1.double())
        got_f = torch.cat([fx[grp].cpu() for grp in range(8)], dim=1).double()
        assert (got_f - f).abs().max().item() < 3e-5 * max(1.0, f.abs().max().item())
        ref = f @ W2.double().T
        got = parts_out.cpu().double().sum(0)
        assert torch.isfinite(got).all()
        assert (got - ref).abs().max().item() < 3e-5 * max(1.0, ref.abs().max().item()), it
