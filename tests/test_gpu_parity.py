"""GPU parity: the HIP path (through the C-ABI) against the CPU oracle and the committed golden vectors.

Tolerances: encoder activations 2e-4 abs (values are O(1..10), fp32 everywhere), logits 1e-3 abs
(BASELINE.json north_star), token ids exact.
"""
import numpy as np
import pytest
import torch

from conftest import GOLDEN_CASES, load_case, sub

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def wt():
    import whisper_trtllm_amd as w
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU is visible")
    w._lib.load()
    return w


def _engines(wt, cfg, weights):
    enc = wt.WhisperEncoderEngine(wt.convert.build_encoder_engine(cfg, weights))
    dec = wt.WhisperDecoderEngine(wt.convert.build_decoder_engine(cfg, weights), cfg)
    return enc, dec


@pytest.mark.parametrize("case", GOLDEN_CASES)
def test_encoder_matches_golden_and_oracle(wt, case):
    import cpu_ref
    z, cfg, weights, mel = load_case(case)
    enc, _ = _engines(wt, cfg, weights)
    out = enc(torch.from_numpy(mel).cuda())
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    assert np.isfinite(got).all()
    np.testing.assert_allclose(sub(got), z["enc_out"], atol=2e-4, rtol=1e-4)
    if cfg["d_model"] <= 192:  # full-tensor check against the oracle where the CPU finishes in seconds
        ref = cpu_ref.encoder_forward(cpu_ref.to_torch(weights), cfg, torch.from_numpy(mel)).numpy()
        np.testing.assert_allclose(got, ref, atol=2e-4, rtol=1e-4)


@pytest.mark.parametrize("case", GOLDEN_CASES)
def test_greedy_ids_and_logits_match_golden(wt, case):
    z, cfg, weights, mel = load_case(case)
    enc, dec = _engines(wt, cfg, weights)
    hidden = enc(torch.from_numpy(mel).cuda())
    B, V, ml = mel.shape[0], cfg["vocab_size"], cfg["max_length"]
    trace = torch.zeros(B, ml - 1, V, dtype=torch.float32, device="cuda")
    ids = dec.generate(hidden, logits_trace=trace).cpu().numpy()
    want = z["ids"]
    assert ids.shape == want.shape, (ids.shape, want.shape)
    steps = want.shape[1] - 1
    L = trace[:, :steps].cpu().numpy()
    stride = int(z["logits_stride"])
    err = np.abs(L[:, :, ::stride] - z["logits_sub"]).max()
    assert err < 1e-3, f"logits differ from the reference by {err}"
    np.testing.assert_allclose(L.max(-1), z["logits_max"], atol=1e-3)
    np.testing.assert_array_equal(ids, want)
