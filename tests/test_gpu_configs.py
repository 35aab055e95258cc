"""GPU: the BASELINE.json configurations at their real model sizes (random-init weights of the real architecture).

config 2  whisper-tiny.en  fp32, batch 1               -> full parity against the CPU oracle
config 3  whisper-small.en fp32, batch 8, KV cache on  -> oracle parity on 2 rows + batch-independence on all 8
config 4  whisper-medium.en fp16 encoder + fp32 decoder, batch 16 -> encoder within 1e-2 of the fp32 oracle's range on
          1 row (parity unpinned: the reference holds no fp16 fixture), decoder logits/ids against the oracle run on the
          ENGINE's fp16 encoder memory (the decoder is fp32: bit-exact ids, logits 1e-3), row independence on all 16
config 5  whisper-medium.en fp32, batch 8 per GPU      -> oracle parity on 1 row + batch-independence on all 8
long      whisper-small.en fp32, batch 8, 128 steps    -> all ids + logits against the oracle (the benchmark's regime:
          hundreds of graph replays, self-cache lengths far beyond the 32-token goldens)
Seeds / mel indices were picked with tests/golden/find_healthy_seeds.py: the oracle's minimum top-2 margin is >= 4e-2 on
every decision asserted here, three orders above the GPU's logit error, so ids are asserted UNCONDITIONALLY (a margin
below 1e-3 fails the test instead of skipping the comparison).
At these sizes the oracle only runs a few rows / steps (seconds of CPU); the rest is covered by properties the domain
offers: utterances are independent, so row b of a batch-8 decode must equal the same utterance decoded alone."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def wt():
    import whisper_trtllm_amd as w
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU is visible")
    w._lib.load()
    return w


def _run(wt, cfg, weights, mel, steps):
    enc = wt.WhisperEncoderEngine(wt.convert.build_encoder_engine(cfg, weights))
    dec = wt.WhisperDecoderEngine(wt.convert.build_decoder_engine(cfg, weights), cfg)
    x = torch.from_numpy(mel).cuda()
    hidden = enc(x)
    B, V = mel.shape[0], cfg["vocab_size"]
    trace = torch.zeros(B, steps, V, dtype=torch.float32, device="cuda")
    ids = dec.generate(hidden, max_length=steps + 1, logits_trace=trace)
    return enc, dec, hidden, trace, ids


def _oracle(cfg, weights, mel, steps):
    import cpu_ref
    W = cpu_ref.to_torch(weights)
    torch.set_num_threads(16)
    with torch.no_grad():
        h = cpu_ref.encoder_forward(W, cfg, torch.from_numpy(mel))
        ids, logits = cpu_ref.greedy_search(W, cfg, h, max_length=steps + 1, return_logits=True)
    return h, ids, logits


def _assert_healthy_margin(logits_ref, floor=1e-3):
    top2 = torch.topk(logits_ref[:, 1:], 2, dim=-1).values      # step 0 emits the forced token whatever the logits
    margin = (top2[..., 0] - top2[..., 1]).min().item()
    assert margin > floor, f"oracle top-2 margin {margin:.2e}: pick another seed (tests/golden/find_healthy_seeds.py)"


@pytest.mark.parametrize("name,batch,oracle_rows,steps", [
    ("whisper-tiny.en", 1, 1, 12),      # config 2
    ("whisper-small.en", 8, 2, 6),      # config 3
    ("whisper-medium.en", 8, 1, 5),     # config 5 (per-GPU shard)
])
def test_baseline_config(wt, name, batch, oracle_rows, steps):
    cfg = wt.synthetic.get_config(name)
    weights = wt.synthetic.make_weights(cfg, 77)
    mel = wt.synthetic.make_mel(cfg, index=300, batch=batch)
    enc, dec, hidden, trace, ids = _run(wt, cfg, weights, mel, steps)
    assert tuple(ids.shape) == (batch, steps + 1) and torch.isfinite(hidden).all() and torch.isfinite(trace).all()
    assert (ids[:, 0] == cfg["decoder_start_token_id"]).all() and (ids[:, 1] == cfg["forced_decoder_ids"][0][1]).all()
    # oracle parity on the first rows
    h_ref, ids_ref, logits_ref = _oracle(cfg, weights, mel[:oracle_rows], steps)
    scale = h_ref.abs().max().item()
    assert (hidden[:oracle_rows].cpu() - h_ref).abs().max().item() < 3e-4 * max(1.0, scale)
    assert (trace[:oracle_rows].cpu() - logits_ref).abs().max().item() < 1e-3
    _assert_healthy_margin(logits_ref)
    np.testing.assert_array_equal(ids[:oracle_rows].cpu().numpy(), ids_ref.numpy())
    # batch independence: every row decoded alone gives the same ids and (to rounding) the same logits
    for b in range(0, batch, max(1, batch // 4)):
        t1 = torch.zeros(1, steps, cfg["vocab_size"], dtype=torch.float32, device="cuda")
        one = dec.generate(enc(torch.from_numpy(mel[b:b + 1]).cuda()), max_length=steps + 1, logits_trace=t1)
        assert (t1[0] - trace[b]).abs().max().item() < 2e-4
        np.testing.assert_array_equal(one.cpu().numpy()[0], ids[b].cpu().numpy())


def test_long_decode_small_en_batch8_matches_oracle(wt):
    """128 decoder steps of whisper-small.en at batch 8 (KV cache on), every row against the oracle: all ids exact, all
    logits within 1e-3.  Oracle minimum margin for this (seed, mel) pair: 7.9e-2 over 8 x 160 decisions."""
    name, batch, steps = "whisper-small.en", 8, 128
    cfg = wt.synthetic.get_config(name)
    weights = wt.synthetic.make_weights(cfg, 77)
    mel = wt.synthetic.make_mel(cfg, index=300, batch=batch)
    enc, dec, hidden, trace, ids = _run(wt, cfg, weights, mel, steps)
    h_ref, ids_ref, logits_ref = _oracle(cfg, weights, mel, steps)
    _assert_healthy_margin(logits_ref)
    assert (hidden.cpu() - h_ref).abs().max().item() < 3e-4 * max(1.0, h_ref.abs().max().item())
    err = (trace.cpu() - logits_ref).abs().amax(dim=(0, 2))           # per step
    assert err.max().item() < 1e-3, f"logits differ by {err.max().item():.2e} at step {int(err.argmax())}"
    np.testing.assert_array_equal(ids.cpu().numpy(), ids_ref.numpy())


def test_headline_model_batch8_64_steps_matches_oracle(wt):
    """The headline benchmark's own model and batch (whisper-medium.en fp32, batch 8) beyond the first few tokens: 64 decoder steps (self
    caches up to 64 rows, 64 replays of the step graph), rows 0 and 1 against the oracle on every id and every logit (about 15 s of host
    CPU), all 8 rows finite.  Oracle minimum top-2 margin of this (seed, mel) pair over the 2 x 64 decisions: 7.5e-3.  The full
    447 steps of all 8 rows stay opt-in (tests/test_gpu_slow.py: minutes of oracle time)."""
    name, batch, steps, rows = "whisper-medium.en", 8, 64, 2
    cfg = wt.synthetic.get_config(name)
    weights = wt.synthetic.make_weights(cfg, 77)
    mel = wt.synthetic.make_mel(cfg, index=300, batch=batch)
    enc, dec, hidden, trace, ids = _run(wt, cfg, weights, mel, steps)
    assert torch.isfinite(trace).all() and tuple(ids.shape) == (batch, steps + 1)
    h_ref, ids_ref, logits_ref = _oracle(cfg, weights, mel[:rows], steps)
    _assert_healthy_margin(logits_ref)
    assert (hidden[:rows].cpu() - h_ref).abs().max().item() < 3e-4 * max(1.0, h_ref.abs().max().item())
    err = (trace[:rows].cpu() - logits_ref).abs().amax(dim=(0, 2))
    assert err.max().item() < 1e-3, f"logits differ by {err.max().item():.2e} at step {int(err.argmax())}"
    np.testing.assert_array_equal(ids[:rows].cpu().numpy(), ids_ref.numpy())


def test_fp16_engines_medium_en_batch16(wt):
    """The size `value_fp16_decoder_b16` is quoted on: whisper-medium.en, fp16 ENCODER and fp16 DECODER engines, batch 16 -- the
    16-row K-split GEMV plan with half weights at d = 1024 / ffn = 4096, half resident K/V caches at H = 16 / S = 1500, the fp16
    cross-K/V projection with its head-split epilogue.  Rows 0 and 1 against the oracle's fp16_engine mode run on the ENGINE's encoder
    memory (fp16-rounded weights and K/V rows, fp32 arithmetic): logits within 2.5e-4 of their range (4e-3 absolute here), ids exact
    (margin asserted: 3.7e-2); row independence on rows 0 / 7 / 15.  The fp16 NUMBERS stay parity-unpinned by nature (the reference holds no fp16 fixture)."""
    import cpu_ref
    cfg = wt.synthetic.get_config("whisper-medium.en")
    weights = wt.synthetic.make_weights(cfg, 77)
    B, steps, V = 16, 5, cfg["vocab_size"]
    mel = wt.synthetic.make_mel(cfg, index=300, batch=B)
    enc16 = wt.WhisperEncoderEngine(wt.convert.build_encoder_engine(cfg, weights, precision="float16"))
    dec16 = wt.WhisperDecoderEngine(wt.convert.build_decoder_engine(cfg, weights, precision="float16"), cfg)
    assert dec16.session.info.precision == 1 and enc16.session.info.precision == 1
    hidden = enc16(torch.from_numpy(mel).cuda())
    trace = torch.zeros(B, steps, V, dtype=torch.float32, device="cuda")
    ids = dec16.generate(hidden, max_length=steps + 1, logits_trace=trace)
    assert torch.isfinite(hidden).all() and torch.isfinite(trace).all() and tuple(ids.shape) == (B, steps + 1)
    torch.set_num_threads(16)
    W16 = cpu_ref.to_torch(cpu_ref.fp16_engine_weights(weights, decoder=True))
    with torch.no_grad():
        ids_ref, logits_ref = cpu_ref.greedy_search(W16, cfg, hidden[:2].cpu(), max_length=steps + 1, return_logits=True, fp16_engine=True)
    err, scale = (trace[:2].cpu() - logits_ref).abs().max().item(), logits_ref.abs().max().item()
    print(f"fp16 engines, medium.en batch 16: logits vs the fp16 oracle {err:.2e} (logit range {scale:.1f})")
    # tests/test_gpu_fp16_decoder.py holds the small models to 2e-3 on logits of range 6..11, i.e. 2-3e-4 of range; medium.en's logits
    # reach 16.6 through 24 layers (fp16 rounding itself moves them by 9e-3 against the fp32 oracle): the same RELATIVE bar here
    # (measured 2.4e-3 = 1.5e-4 of range; the remainder is the folded cross-query matrix, rounded to fp16 once more as a matrix)
    assert err < 2.5e-4 * max(scale, 8.0), (err, scale)
    top2 = torch.topk(logits_ref[:, 1:], 2, dim=-1).values
    margin = (top2[..., 0] - top2[..., 1]).min().item()
    assert margin > 8e-3, f"fp16 oracle top-2 margin {margin:.2e} too thin for an id comparison: pick another seed"
    np.testing.assert_array_equal(ids[:2].cpu().numpy(), ids_ref.numpy())
    for b in (0, 7, 15):
        t1 = torch.zeros(1, steps, V, dtype=torch.float32, device="cuda")
        one = dec16.generate(enc16(torch.from_numpy(mel[b:b + 1]).cuda()), max_length=steps + 1, logits_trace=t1)
        assert (t1[0] - trace[b]).abs().max().item() < 4e-3   # the fp16 ENCODER's GEMM tiles see other rows' positions: not bit-equal
        np.testing.assert_array_equal(one.cpu().numpy()[0], ids[b].cpu().numpy())


def test_config4_fp16_encoder_fp32_decoder_batch16(wt):
    """BASELINE config 4 at its real size: whisper-medium.en, fp16 encoder engine + fp32 decoder engine, batch 16.
    (a) fp16 encoder memory of row 0 within 2e-3 of the dynamic range of the oracle's fp16_engine mode (fp16-rounded weights and
        GEMM-input activations, fp32 arithmetic: the engine's arithmetic restated) and within 1e-2 of the fp32 oracle -- the fp16
        NUMBERS stay parity-unpinned: the reference publishes no fp16 result (README.md:82-88);
    (b) the fp32 decoder on that memory against the oracle's decoder on the SAME memory: logits 1e-3, ids exact;
    (c) utterances are independent: rows decoded alone give the same ids as in the batch of 16."""
    import cpu_ref
    cfg = wt.synthetic.get_config("whisper-medium.en")
    weights = wt.synthetic.make_weights(cfg, 77)
    B, steps, V = 16, 5, cfg["vocab_size"]
    mel = wt.synthetic.make_mel(cfg, index=300, batch=B)
    enc16 = wt.WhisperEncoderEngine(wt.convert.build_encoder_engine(cfg, weights, precision="float16"))
    dec = wt.WhisperDecoderEngine(wt.convert.build_decoder_engine(cfg, weights), cfg)
    hidden = enc16(torch.from_numpy(mel).cuda())
    trace = torch.zeros(B, steps, V, dtype=torch.float32, device="cuda")
    ids = dec.generate(hidden, max_length=steps + 1, logits_trace=trace)
    assert torch.isfinite(hidden).all() and torch.isfinite(trace).all() and tuple(ids.shape) == (B, steps + 1)
    W = cpu_ref.to_torch(weights)
    torch.set_num_threads(16)
    with torch.no_grad():
        h_ref = cpu_ref.encoder_forward(W, cfg, torch.from_numpy(mel[:1]))
        scale = h_ref.abs().max().item()
        err = (hidden[:1].cpu() - h_ref).abs().max().item()
        assert err < 1e-2 * scale, (err, scale)
        W16 = cpu_ref.to_torch(cpu_ref.fp16_engine_weights(weights, encoder=True))
        h_ref16 = cpu_ref.encoder_forward(W16, cfg, torch.from_numpy(mel[:1]), fp16_engine=True)
        err16 = (hidden[:1].cpu() - h_ref16).abs().max().item()
        print(f"config 4 row 0: fp16 encoder vs fp16 oracle {err16 / scale:.2e} of range, vs fp32 oracle {err / scale:.2e}")
        assert err16 < 2e-3 * scale, (err16, scale)
        del W16
        ids_ref, logits_ref = cpu_ref.greedy_search(W, cfg, hidden[:2].cpu(), max_length=steps + 1, return_logits=True)
    _assert_healthy_margin(logits_ref)
    assert (trace[:2].cpu() - logits_ref).abs().max().item() < 1e-3
    np.testing.assert_array_equal(ids[:2].cpu().numpy(), ids_ref.numpy())
    for b in (0, 5, 10, 15):
        t1 = torch.zeros(1, steps, V, dtype=torch.float32, device="cuda")
        one = dec.generate(enc16(torch.from_numpy(mel[b:b + 1]).cuda()), max_length=steps + 1, logits_trace=t1)
        assert (t1[0] - trace[b]).abs().max().item() < 2e-3   # fp16 GEMM tiles see other rows' positions: not bit-equal
        np.testing.assert_array_equal(one.cpu().numpy()[0], ids[b].cpu().numpy())
