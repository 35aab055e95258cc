"""CPU: the oracle (oracle/cpu_ref.py) against the golden vectors recorded from the reference's bundled HF
Whisper (tests/golden/make_golden.py).  Tolerances: activations/logits 1e-4 abs (same torch ops, only the
thread-count-dependent summation order differs), token ids exact."""
import numpy as np
import pytest
import torch

import cpu_ref
from conftest import GOLDEN_CASES, load_case, sub

FAST = [c for c in GOLDEN_CASES if not c.startswith("tiny")]


@pytest.mark.parametrize("case", GOLDEN_CASES)
def test_encoder_matches_reference(case):
    z, cfg, weights, mel = load_case(case)
    W = cpu_ref.to_torch(weights)
    x = torch.from_numpy(mel)
    with torch.no_grad():
        h0 = cpu_ref.encoder_conv_frontend(W, cfg, x)
        h1 = cpu_ref.encoder_layer(W, cfg, 0, h0)
        out = cpu_ref.encoder_forward(W, cfg, x)
    np.testing.assert_allclose(sub(h0.numpy()), z["frontend"], atol=1e-4)
    np.testing.assert_allclose(sub(h1.numpy()), z["enc_layer0"], atol=1e-4)
    np.testing.assert_allclose(sub(out.numpy()), z["enc_out"], atol=1e-4)
    assert abs(float(out.abs().max()) - float(z["enc_out_absmax"])) < 1e-3


@pytest.mark.parametrize("case", GOLDEN_CASES)
def test_greedy_ids_and_logits_match_reference(case):
    z, cfg, weights, mel = load_case(case)
    W = cpu_ref.to_torch(weights)
    with torch.no_grad():
        enc = cpu_ref.encoder_forward(W, cfg, torch.from_numpy(mel))
        ids, logits = cpu_ref.greedy_search(W, cfg, enc, return_logits=True)
    np.testing.assert_array_equal(ids.numpy(), z["ids"])
    L = logits.numpy()
    stride = int(z["logits_stride"])
    np.testing.assert_allclose(L[:, :, ::stride], z["logits_sub"], atol=1e-4)
    np.testing.assert_array_equal(L.argmax(-1), z["logits_argmax"])
    assert float(z["logits_margin"].min()) > 1e-3, "fixture has a near-tie; regenerate with another seed"


@pytest.mark.parametrize("case", FAST)
def test_kv_rows_match_reference(case):
    z, cfg, weights, mel = load_case(case)
    W = cpu_ref.to_torch(weights)
    ids = torch.from_numpy(z["ids"])
    with torch.no_grad():
        enc = cpu_ref.encoder_forward(W, cfg, torch.from_numpy(mel))
        past = None
        for t in range(ids.shape[1] - 1):
            _, past = cpu_ref.decoder_forward(W, cfg, ids[:, t:t + 1], enc, past)
            np.testing.assert_allclose(past[0][0][:, :, -1].numpy(), z["self_k0_rows"][:, t], atol=1e-4)
            np.testing.assert_allclose(past[-1][1][:, :, -1].numpy(), z["self_vL_rows"][:, t], atol=1e-4)
    np.testing.assert_allclose(sub(past[0][2].numpy()), z["cross_k0"], atol=1e-4)
    np.testing.assert_allclose(sub(past[-1][3].numpy()), z["cross_vL"], atol=1e-4)


@pytest.mark.parametrize("case", ["toy-short_b3", "toy-wide_b2"])
def test_engine_surface_equals_hf_protocol(case):
    """The TRT-LLM engine contract (mask-length gated caches, model.py:261-281) reproduces the HF cache
    protocol when driven the way run.py:109-126 drives it: step 0 with length-1 masks and dummy caches,
    then masks of length 1+k / S+1."""
    z, cfg, weights, mel = load_case(case)
    W = cpu_ref.to_torch(weights)
    L, H, S = cfg["decoder_layers"], cfg["decoder_attention_heads"], cfg["max_source_positions"]
    ids = torch.from_numpy(z["ids"])
    with torch.no_grad():
        enc = cpu_ref.encoder_forward(W, cfg, torch.from_numpy(mel))[:1]
        past = None
        g = torch.Generator().manual_seed(0)
        sk, sv = torch.rand(L, H, 1, 64, generator=g), torch.rand(L, H, 1, 64, generator=g)
        ck, cv = torch.rand(L, H, S, 64, generator=g), torch.rand(L, H, S, 64, generator=g)
        m_s, m_c = 1, 1
        for t in range(min(6, ids.shape[1] - 1)):
            ref_logits, past = cpu_ref.decoder_forward(W, cfg, ids[:1, t:t + 1], enc, past)
            lg, sk, sv, ck, cv = cpu_ref.engine_decoder_step(W, cfg, ids[:1, t:t + 1].int(), enc, sk, sv, ck, cv, m_s, m_c)
            assert sk.shape == (L, H, t + 1, 64) and ck.shape == (L, H, S, 64)
            np.testing.assert_allclose(lg.numpy(), ref_logits.numpy(), atol=1e-4)
            np.testing.assert_allclose(sk[0].numpy(), past[0][0][0].numpy(), atol=1e-5)
            np.testing.assert_allclose(cv[-1].numpy(), past[-1][3][0].numpy(), atol=1e-5)
            m_s, m_c = 1 + sk.shape[2], S + 1


def test_engine_surface_partial_cross_cache():
    """Intermediate cross-cache lengths: cur = proj(enc[0 : S-c]) is concatenated BEHIND past[:c]
    (the slice starts at 0, model.py:265-266) — only c in {0, S} is meaningful, but the contract is defined."""
    import whisper_trtllm_amd  # noqa: F401
    from whisper_trtllm_amd import synthetic
    cfg = synthetic.get_config("toy-short")
    W = cpu_ref.to_torch(synthetic.make_weights(cfg, 5))
    L, H, S, d = cfg["decoder_layers"], cfg["decoder_attention_heads"], cfg["max_source_positions"], cfg["d_model"]
    g = torch.Generator().manual_seed(1)
    enc = torch.randn(1, S, d, generator=g)
    sk, sv = torch.randn(L, H, 3, 64, generator=g), torch.randn(L, H, 3, 64, generator=g)
    ck, cv = torch.randn(L, H, S, 64, generator=g), torch.randn(L, H, S, 64, generator=g)
    with torch.no_grad():
        lg, nsk, nsv, nck, ncv = cpu_ref.engine_decoder_step(W, cfg, torch.tensor([[7]], dtype=torch.int32), enc, sk, sv, ck, cv, 3, 11)
    assert nsk.shape == (L, H, 3, 64) and nck.shape == (L, H, S, 64)   # self cache_len = min(3-1, 3) = 2 -> 3 rows
    torch.testing.assert_close(nsk[:, :, :2], sk[:, :, :2])
    torch.testing.assert_close(nck[:, :, :10], ck[:, :, :10])
    k0 = torch.nn.functional.linear(enc[:, :S - 10], W["model.decoder.layers.0.encoder_attn.k_proj.weight"])
    torch.testing.assert_close(nck[0, :, 10:], k0.view(1, S - 10, H, 64).transpose(1, 2)[0], atol=1e-5, rtol=1e-5)
    assert torch.isfinite(lg).all()


def test_fp16_engine_modes_are_small_perturbations_of_the_pinned_fp32_oracle():
    """The oracle's fp16_engine modes (the checker of the fp16 engines, tests/test_gpu_fp16_decoder.py / test_fp16_encoder_engine) are the
    pinned fp32 restatement with roundings inserted: only weight MATRICES are rounded (biases, LayerNorm parameters and position tables
    stay fp32, the tied vocabulary matrix stays tied), the outputs stay within fp16 rounding of the fp32 path, and with fp16_engine off
    the functions are bit-for-bit the fp32 oracle the goldens pin."""
    z, cfg, weights, mel = load_case("toy-short_b3")
    w16 = cpu_ref.fp16_engine_weights(weights, encoder=True, decoder=True)
    for k, v in weights.items():
        is_matrix = k.endswith(".weight") and v.ndim >= 2 and "embed_positions" not in k
        if is_matrix:
            assert np.array_equal(w16[k], v.astype(np.float16).astype(np.float32)), k
        else:
            assert w16[k] is v or np.array_equal(w16[k], v), k
    assert w16["proj_out.weight"] is w16["model.decoder.embed_tokens.weight"]
    only_dec = cpu_ref.fp16_engine_weights(weights, decoder=True)
    assert only_dec["model.encoder.layers.0.fc1.weight"] is weights["model.encoder.layers.0.fc1.weight"]
    W, W16 = cpu_ref.to_torch(weights), cpu_ref.to_torch(w16)
    x = torch.from_numpy(mel)
    with torch.no_grad():
        h32 = cpu_ref.encoder_forward(W, cfg, x)
        assert torch.equal(h32, cpu_ref.encoder_forward(W, cfg, x, fp16_engine=False))
        h16 = cpu_ref.encoder_forward(W16, cfg, x, fp16_engine=True)
        scale = h32.abs().max().item()
        err = (h16 - h32).abs().max().item()
        assert 1e-5 * scale < err < 1e-2 * scale, (err, scale)           # rounded, but only at fp16 precision
        ids32, lg32 = cpu_ref.greedy_search(W, cfg, h32, return_logits=True)
        ids16, lg16 = cpu_ref.greedy_search(W16, cfg, h32, return_logits=True, fp16_engine=True)
        n = min(lg32.shape[1], lg16.shape[1])
        assert (lg16[:, :n] - lg32[:, :n]).abs().max().item() < 5e-2 * lg32.abs().max().item()
    np.testing.assert_array_equal(ids32.numpy(), z["ids"])
