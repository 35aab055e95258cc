"""GPU: the reference's Session surface (by-value caches, mask-length gating, tensor names) through
wt_engine_infer_shapes / wt_engine_run, against the oracle's engine-surface restatement, and the run.py flow
(engine wrappers + Python greedy loop) against the golden ids."""
import importlib.util
import os
import pickle
import sys
import types

import numpy as np
import pytest
import torch

from conftest import load_case

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def wt():
    import whisper_trtllm_amd as w
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU is visible")
    w._lib.load()
    return w


def _dec_inputs(wt, cfg, tok, enc, sk, sv, ck, cv, m_s, m_c):
    return {"data": torch.tensor([[tok]], dtype=torch.int32, device="cuda"),
            "length": torch.tensor([1], dtype=torch.int32, device="cuda"),
            "encoder_hidden_states": enc.cuda(), "self_past_key": sk.cuda(), "self_past_value": sv.cuda(),
            "cross_past_key": ck.cuda(), "cross_past_value": cv.cuda(),
            "past_self_cache_mask": torch.rand(m_s, device="cuda"), "past_cross_cache_mask": torch.rand(m_c, device="cuda")}


@pytest.mark.parametrize("case", ["toy-short_b3", "toy-wide_b2"])
def test_decoder_session_protocol_matches_oracle(wt, case):
    import cpu_ref
    z, cfg, weights, mel = load_case(case)
    W = cpu_ref.to_torch(weights)
    L, H, S, V = cfg["decoder_layers"], cfg["decoder_attention_heads"], cfg["max_source_positions"], cfg["vocab_size"]
    sess = wt.Session.from_serialized_engine(wt.convert.build_decoder_engine(cfg, weights))
    ids = z["ids"]
    with torch.no_grad():
        enc = cpu_ref.encoder_forward(W, cfg, torch.from_numpy(mel))[:1]
    g = torch.Generator().manual_seed(3)
    sk, sv = torch.rand(L, H, 1, 64, generator=g), torch.rand(L, H, 1, 64, generator=g)      # run.py:114-117 dummies
    ck, cv = torch.rand(L, H, S, 64, generator=g), torch.rand(L, H, S, 64, generator=g)
    m_s, m_c = 1, 1
    for t in range(5):
        tok = int(ids[0, t])
        with torch.no_grad():
            want = cpu_ref.engine_decoder_step(W, cfg, torch.tensor([[tok]], dtype=torch.int32), enc, sk, sv, ck, cv, m_s, m_c)
        out = sess._debug_run(_dec_inputs(wt, cfg, tok, enc, sk, sv, ck, cv, m_s, m_c))
        assert tuple(out["hidden_states"].shape) == (1, 1, V)
        assert tuple(out["next_self_keys"].shape) == (L, H, t + 1, 64) and tuple(out["next_cross_values"].shape) == (L, H, S, 64)
        names = ["hidden_states", "next_self_keys", "next_self_values", "next_cross_keys", "next_cross_values"]
        for n, w_ in zip(names, want):
            err = (out[n].cpu() - w_.reshape(out[n].shape)).abs().max().item()
            assert err < (1e-3 if n == "hidden_states" else 1e-4), (t, n, err)
        sk, sv, ck, cv = (out[n].cpu() for n in names[1:])
        m_s, m_c = 1 + sk.shape[2], S + 1


def test_decoder_session_gating_edge_cases(wt):
    """self cache_len = min(m_s-1, s) with an over-long past; partial cross cache (model.py:264-272, :278)."""
    import cpu_ref
    cfg = wt.synthetic.get_config("toy-short")
    weights = wt.synthetic.make_weights(cfg, 5)
    W = cpu_ref.to_torch(weights)
    L, H, S, d = cfg["decoder_layers"], cfg["decoder_attention_heads"], cfg["max_source_positions"], cfg["d_model"]
    sess = wt.Session.from_serialized_engine(wt.convert.build_decoder_engine(cfg, weights))
    g = torch.Generator().manual_seed(1)
    enc = torch.randn(1, S, d, generator=g)
    for (s, m_s, m_c) in ((3, 3, 11), (6, 3, S + 1), (2, 5, 1), (1, 1, S)):
        sk, sv = torch.randn(L, H, s, 64, generator=g), torch.randn(L, H, s, 64, generator=g)
        ck, cv = torch.randn(L, H, S, 64, generator=g), torch.randn(L, H, S, 64, generator=g)
        with torch.no_grad():
            want = cpu_ref.engine_decoder_step(W, cfg, torch.tensor([[9]], dtype=torch.int32), enc, sk, sv, ck, cv, m_s, m_c)
        out = sess._debug_run(_dec_inputs(wt, cfg, 9, enc, sk, sv, ck, cv, m_s, m_c))
        for n, w_ in zip(["hidden_states", "next_self_keys", "next_self_values", "next_cross_keys", "next_cross_values"], want):
            assert tuple(out[n].shape) == tuple(w_.reshape(out[n].shape).shape)
            assert (out[n].cpu() - w_.reshape(out[n].shape)).abs().max().item() < 1e-3, (s, m_s, m_c, n)


def test_infer_shapes_error_conventions(wt):
    """Unknown name / wrong dtype -> logged error + None (session.py:131-136); run before infer_shapes -> False."""
    cfg = wt.synthetic.get_config("toy-short")
    weights = wt.synthetic.make_weights(cfg, 5)
    TI, trt = wt.TensorInfo, wt.trt
    enc = wt.Session.from_serialized_engine(wt.convert.build_encoder_engine(cfg, weights))
    F = 2 * cfg["max_source_positions"]
    assert enc.run({"data": torch.zeros(1, 80, F).cuda()}, {"hidden_states": torch.zeros(1).cuda()}, 0) is False
    assert enc.infer_shapes([TI("data", trt.float16, (1, 80, F)), TI("length", trt.float32, (1,))]) is None
    assert enc.infer_shapes([TI("datum", trt.float32, (1, 80, F))]) is None
    assert enc.infer_shapes([TI("data", trt.float32, (1, 80, F + 2)), TI("length", trt.float32, (1,))]) is None
    out = enc.infer_shapes([TI("data", trt.float32, (3, 80, F)), TI("length", trt.float32, (3,))])
    assert [(o.name, o.dtype, tuple(o.shape)) for o in out] == [("hidden_states", trt.float32, (3, cfg["max_source_positions"], cfg["d_model"]))]
    dec = wt.Session.from_serialized_engine(wt.convert.build_decoder_engine(cfg, weights))
    assert dec.infer_shapes([TI("data", trt.float32, (1, 1))]) is None                       # wrong dtype
    assert dec.infer_shapes([TI("data", trt.int32, (2, 1))]) is None                          # batch is fixed to 1
    with pytest.raises(RuntimeError):
        wt.Session.from_serialized_engine(b"garbage" * 100)
    with pytest.raises(ValueError):
        wt.WhisperEncoderEngine(wt.convert.build_decoder_engine(cfg, weights))


def test_run_py_flow_matches_golden(wt, tmp_path):
    """examples/whisper: build_* artefacts on disk -> run.py wrappers + greedy_search (Session path) and the fast path."""
    z, cfg, weights, mel = load_case("toy-short-eos1_b3")
    eng = tmp_path / "eng"
    eng.mkdir()
    (eng / "WhisperEncoder.engine").write_bytes(wt.convert.build_encoder_engine(cfg, weights))
    (eng / "WhisperDecoder.engine").write_bytes(wt.convert.build_decoder_engine(cfg, weights))
    (eng / "config.pkl").write_bytes(pickle.dumps(cfg))
    sys.path.insert(0, os.path.join(ROOT, "examples", "whisper"))
    spec = importlib.util.spec_from_file_location("wt_example_run", os.path.join(ROOT, "examples", "whisper", "run.py"))
    run = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(run)
    args = types.SimpleNamespace(engine_dir=str(eng))
    config = pickle.loads((eng / "config.pkl").read_bytes())
    we, wd = run.WhisperEncoder(args, config), run.WhisperDecoder(args, config)
    # batch-1 semantics of the reference: a finished row stops its own loop, so compare up to (and including) EOS
    for b in range(mel.shape[0]):
        got = run.decode_with_sessions(we, wd, config, torch.from_numpy(mel[b:b + 1]).cuda())[0].cpu().numpy()
        want = z["ids"][b]
        stop = np.where(want[1:] == cfg["eos_token_id"])[0]
        n = (stop[0] + 2) if len(stop) else len(want)
        np.testing.assert_array_equal(got, want[:n])


def test_fast_path_argument_errors(wt):
    """The C-ABI returns negative codes + a message (never throws); the shim raises RuntimeError with that message."""
    import ctypes
    cfg = wt.synthetic.get_config("toy-short")
    weights = wt.synthetic.make_weights(cfg, 5)
    enc = wt.WhisperEncoderEngine(wt.convert.build_encoder_engine(cfg, weights))
    dec = wt.WhisperDecoderEngine(wt.convert.build_decoder_engine(cfg, weights), cfg)
    hidden = enc(torch.from_numpy(wt.synthetic.make_mel(cfg, 0, 2)).cuda())
    with pytest.raises(ValueError):
        enc(torch.zeros(1, 80, 100, device="cuda"))                         # wrong frame count
    with pytest.raises(ValueError):
        dec.begin(hidden[:, :10])                                           # wrong encoder memory shape
    with pytest.raises(ValueError, match="CUDA"):
        enc(torch.from_numpy(wt.synthetic.make_mel(cfg, 0, 2)))             # a HOST tensor must never reach the C-ABI as a device pointer
    with pytest.raises(ValueError, match="CUDA"):
        dec.begin(hidden.cpu())
    with pytest.raises(RuntimeError, match="max_length"):
        dec.begin(hidden, max_length=cfg["max_target_positions"] + 5)       # beyond max_target_positions
    bad = dict(cfg)
    bad["suppress_tokens"] = [cfg["vocab_size"] + 3]
    with pytest.raises(RuntimeError, match="outside the vocabulary"):
        wt.WhisperDecoderEngine(wt.convert.build_decoder_engine(cfg, weights), bad).begin(hidden)
    lib = wt._lib.load()
    nine = torch.zeros(17, cfg["max_source_positions"], cfg["d_model"], device="cuda")
    dec.begin(hidden)
    rc = lib.wt_decoder_begin(dec.session.handle, nine.data_ptr(), 17, ctypes.byref(dec._p), None)
    assert rc == -38 and "shard the batch" in wt._lib.last_error()          # WT_E_UNSUPPORTED: > 16 utterances per call
    assert lib.wt_encoder_forward(dec.session.handle, nine.data_ptr(), 1, nine.data_ptr(), None) == -22   # decoder handle
    # the engine still works after the failed calls
    ids = dec.generate(hidden)
    assert ids.shape[0] == 2


def test_c_abi_greedy_convenience_call(wt):
    """wt_decoder_greedy (begin + steps/poll loop inside the library) returns the golden ids and the stop length."""
    import ctypes
    z, cfg, weights, mel = load_case("toy-short-eos1_b3")
    enc = wt.WhisperEncoderEngine(wt.convert.build_encoder_engine(cfg, weights))
    dec = wt.WhisperDecoderEngine(wt.convert.build_decoder_engine(cfg, weights), cfg)
    hidden = enc(torch.from_numpy(mel).cuda()).contiguous()
    p = dec._params(None, None, None)
    ml = cfg["max_length"]
    ids = torch.zeros(3, ml, dtype=torch.int32, device="cuda")
    out_len = ctypes.c_int()
    lib = wt._lib.load()
    lib.wt_decoder_greedy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(wt._lib.GreedyParams),
                                      ctypes.c_void_p, ctypes.POINTER(ctypes.c_int), ctypes.c_void_p]
    rc = lib.wt_decoder_greedy(dec.session.handle, hidden.data_ptr(), 3, ctypes.byref(p), ids.data_ptr(), ctypes.byref(out_len), None)
    assert rc == 0, wt._lib.last_error()
    assert out_len.value == z["ids"].shape[1]
    np.testing.assert_array_equal(ids[:, :out_len.value].cpu().numpy(), z["ids"])


def _toy_vocabulary(cfg, ckpt):
    """vocab.json for a toy config: the 256 byte symbols, then " <letter><letter>" style tokens, <|endoftext|> at the config's eos id."""
    import json
    from whisper_trtllm_amd.text import _byte_decoder
    symbols = sorted(_byte_decoder(), key=_byte_decoder().get)
    space = symbols[32]
    vocab = {s: i for i, s in enumerate(symbols)}
    letters = "etaoinshrdlucmfwyp"
    k = 0
    while len(vocab) < cfg["vocab_size"]:
        tok = space + letters[k % len(letters)] + letters[(k // len(letters)) % len(letters)] + ("s" if k >= len(letters) ** 2 else "")
        if len(vocab) == cfg["eos_token_id"]:
            tok = "<|endoftext|>"
        vocab.setdefault(tok, len(vocab))
        k += 1
    json.dump(vocab, open(ckpt / "vocab.json", "w", encoding="utf-8"), ensure_ascii=False)


@pytest.mark.parametrize("batching", ["continuous", "sorted"])
def test_cal_wer_script_end_to_end(wt, tmp_path, batching):
    """examples/whisper/cal_wer.py as a subprocess over artefacts this test writes itself: toy engines + config.pkl, a toy byte-level
    vocabulary, and a `librispeech.cache` of (log-mel, reference text) pairs — fast-path decode, token decode, English normaliser,
    pooled WER.  The expected number is recomputed here from the same library calls.  Both plans -- continuous (the default: dataset
    order, slots refilled on the device) and length-sorted batches -- must give the hypotheses of the reference's one-clip-at-a-time
    loop: every row ends AT its EOS (this golden's pad token is an ordinary text token, so a row that padded on would show)."""
    import json
    import subprocess
    from whisper_trtllm_amd.english import EnglishTextNormalizer
    from whisper_trtllm_amd.text import WhisperTokenDecoder, word_error_rate
    z, cfg, weights, mel = load_case("toy-short-eos1_b3")
    eng, ckpt = tmp_path / "eng", tmp_path / "ckpt"
    eng.mkdir()
    ckpt.mkdir()
    (eng / "WhisperEncoder.engine").write_bytes(wt.convert.build_encoder_engine(cfg, weights))
    (eng / "WhisperDecoder.engine").write_bytes(wt.convert.build_decoder_engine(cfg, weights))
    (eng / "config.pkl").write_bytes(pickle.dumps(cfg))
    _toy_vocabulary(cfg, ckpt)
    json.dump({"colour": "color"}, open(ckpt / "normalizer.json", "w"))
    enc = wt.WhisperEncoderEngine((eng / "WhisperEncoder.engine").read_bytes())
    dec = wt.WhisperDecoderEngine((eng / "WhisperDecoder.engine").read_bytes(), cfg)
    ids = dec.generate(enc(torch.from_numpy(mel).cuda())).cpu().tolist()
    eos = cfg["eos_token_id"]
    assert cfg["pad_token_id"] != eos and any(eos in r[1:-1] for r in ids)      # a row of this golden stops early and pads on in a batch
    ids = [r[:r.index(eos, 1) + 1] if eos in r[1:] else r for r in ids]          # run.py:219-226: a clip decoded alone ends at its EOS
    hyp = WhisperTokenDecoder.from_dir(str(ckpt)).batch_decode(ids, skip_special_tokens=True)
    refs = [hyp[0], "twenty one colour " + hyp[1], "completely different words here"][:len(hyp)]
    pickle.dump([(mel[i], refs[i]) for i in range(len(refs))], open(tmp_path / "librispeech.cache", "wb"))
    norm = EnglishTextNormalizer({"colour": "color"})
    want = word_error_rate([norm(t) for t in refs], [norm(t) for t in hyp])
    out = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "whisper", "cal_wer.py"), "--whisper", str(ckpt), "--engine_dir", str(eng),
                          "--cache", str(tmp_path / "librispeech.cache"), "--batch", "2", "--batching", batching], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("WER:")][-1]
    assert abs(float(line.split()[1]) - want * 100) < 0.006, (line, want)
    assert 0.0 < want < 10.0


def test_run_py_transcribes_audio_files(wt, tmp_path):
    """examples/whisper/run.py --audio: .wav -> GPU log-mel front-end -> engines -> ids -> text, as a subprocess over artefacts written
    here; the expected transcripts are recomputed through the same library calls."""
    import subprocess
    import wave
    from whisper_trtllm_amd.audio import LogMelFrontend
    from whisper_trtllm_amd.text import WhisperTokenDecoder
    cfg = wt.synthetic.get_config("toy")          # 30 s inputs (max_source_positions 1500), 512-token vocabulary
    weights = wt.synthetic.make_weights(cfg, 9)
    eng, ckpt = tmp_path / "eng", tmp_path / "ckpt"
    eng.mkdir()
    ckpt.mkdir()
    (eng / "WhisperEncoder.engine").write_bytes(wt.convert.build_encoder_engine(cfg, weights))
    (eng / "WhisperDecoder.engine").write_bytes(wt.convert.build_decoder_engine(cfg, weights))
    (eng / "config.pkl").write_bytes(pickle.dumps(cfg))
    _toy_vocabulary(cfg, ckpt)
    rng = np.random.default_rng(11)
    paths, waves = [], []
    for k, seconds in enumerate((1.5, 0.8)):
        t = np.arange(int(16000 * seconds)) / 16000.0
        pcm = np.clip(np.round((0.4 * np.sin(2 * np.pi * (300 + 170 * k) * t) + 0.03 * rng.standard_normal(t.size)) * 32768.0), -32768, 32767).astype("<i2")
        path = tmp_path / f"clip{k}.wav"
        with wave.open(str(path), "wb") as f:
            f.setnchannels(1)
            f.setsampwidth(2)
            f.setframerate(16000)
            f.writeframes(pcm.tobytes())
        paths.append(str(path))
        w30 = np.zeros(480000, dtype=np.float32)
        w30[:pcm.size] = pcm.astype(np.float32) / 32768.0
        waves.append(w30)
    mel = LogMelFrontend()(torch.from_numpy(np.stack(waves)).cuda())
    enc = wt.WhisperEncoderEngine((eng / "WhisperEncoder.engine").read_bytes())
    dec = wt.WhisperDecoderEngine((eng / "WhisperDecoder.engine").read_bytes(), cfg)
    want = WhisperTokenDecoder.from_dir(str(ckpt)).batch_decode(dec.generate(enc(mel)).cpu().tolist(), skip_special_tokens=True)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "whisper", "run.py"), "--engine_dir", str(eng), "--whisper", str(ckpt),
                          "--audio"] + paths, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    for k, text in enumerate(want):
        assert f"clip{k}.wav: {text!r}" in out.stdout, out.stdout[-1500:]


def test_plain_c_host_of_the_c_abi(wt, tmp_path):
    """examples/c/wt_greedy: a C/C++ program that uses ONLY include/whisper_trtllm_amd.h + the HIP runtime (no Python, no torch in
    the process) reads engine files, a raw float32 log-mel file and the token rules, and prints the same ids as the golden file."""
    import subprocess
    exe = os.path.join(ROOT, "examples", "c", "wt_greedy")
    if not os.path.exists(exe):   # normally built by __graft_entry__.build(); hipcc is in the image
        spec = importlib.util.spec_from_file_location("_wt_build", os.path.join(ROOT, "whisper-trtllm_amd", "build.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        mod.build_c_example()
    z, cfg, weights, mel = load_case("toy-short-eos1_b3")
    (tmp_path / "enc.engine").write_bytes(wt.convert.build_encoder_engine(cfg, weights))
    (tmp_path / "dec.engine").write_bytes(wt.convert.build_decoder_engine(cfg, weights))
    np.ascontiguousarray(mel, dtype=np.float32).tofile(tmp_path / "mel.f32")
    forced = cfg.get("forced_decoder_ids") or []
    begin_index = (1 if cfg.get("forced_bos_token_id") is None else 2) + (forced[-1][0] if forced else 0)
    rules = [cfg["decoder_start_token_id"], cfg["eos_token_id"], cfg["pad_token_id"], cfg["max_length"], begin_index]
    for lst in (cfg.get("suppress_tokens") or [], cfg.get("begin_suppress_tokens") or []):
        rules += [len(lst)] + list(lst)
    rules += [len(forced)] + [x for pair in forced for x in pair]
    (tmp_path / "rules.txt").write_text(" ".join(str(int(x)) for x in rules))
    out = subprocess.run([exe, str(tmp_path / "enc.engine"), str(tmp_path / "dec.engine"), str(tmp_path / "mel.f32"), str(mel.shape[0]),
                          str(tmp_path / "rules.txt")], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    got = np.array([[int(t) for t in line.split()] for line in out.stdout.strip().splitlines()])
    want = z["ids"]
    np.testing.assert_array_equal(got, want[:, :got.shape[1]])
    assert got.shape[0] == want.shape[0] and (want[:, got.shape[1]:] == cfg["pad_token_id"]).all()
    # the same program with 4 host threads on clones of the two engines (wt_engine_clone, std::thread, one stream each): it exits
    # non-zero unless every worker decoded the same ids
    out4 = subprocess.run([exe, str(tmp_path / "enc.engine"), str(tmp_path / "dec.engine"), str(tmp_path / "mel.f32"), str(mel.shape[0]),
                           str(tmp_path / "rules.txt"), "4"], capture_output=True, text=True, timeout=300)
    assert out4.returncode == 0, out4.stderr[-2000:]
    assert out4.stdout == out.stdout
    # ... and through the continuous mode of the C-ABI (wt_decoder_stream_*, 2 slots for 3 utterances): every row ends at its own EOS
    outs = subprocess.run([exe, str(tmp_path / "enc.engine"), str(tmp_path / "dec.engine"), str(tmp_path / "mel.f32"), str(mel.shape[0]),
                           str(tmp_path / "rules.txt"), "stream2"], capture_output=True, text=True, timeout=300)
    assert outs.returncode == 0, outs.stderr[-2000:]
    rows = [[int(t) for t in line.split()] for line in outs.stdout.strip().splitlines()]
    eos = cfg["eos_token_id"]
    expect = [[int(t) for t in (list(r)[:list(r).index(eos, 1) + 1] if eos in list(r)[1:] else r)] for r in want]
    assert rows == expect


def _write_engine_dir(wt, tmp_path, cfg, weights):
    eng = tmp_path / "eng"
    eng.mkdir()
    (eng / "WhisperEncoder.engine").write_bytes(wt.convert.build_encoder_engine(cfg, weights))
    (eng / "WhisperDecoder.engine").write_bytes(wt.convert.build_decoder_engine(cfg, weights))
    (eng / "config.pkl").write_bytes(pickle.dumps(cfg))
    return eng


def _torchrun(nproc, script_args, timeout=600):
    import subprocess
    port = 29600 + os.getpid() % 300
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
           "--master-port", str(port)] + script_args
    return subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env)


def test_two_ranks_decode_their_shards_through_the_hip_engine(wt, tmp_path):
    """N > 1 on real hardware: two fresh processes (one per rank, both on this box's GPU -- LOCAL_RANK modulo the visible devices --
    gloo for the barrier / max / gather, exactly the calls bench.py and the scripts make) each decode their `utterance_shard`
    through WhisperEncoderEngine / WhisperDecoderEngine; the gathered ids must equal the single-process decode.  5 utterances over
    2 ranks = shards of 3 and 2 (ragged); config 5's driver is this script under --nproc-per-node 8."""
    import json
    z, cfg, weights, mel = load_case("toy-short-eos1_b3")        # rows finish at different lengths (pad != eos)
    eng = _write_engine_dir(wt, tmp_path, cfg, weights)
    n, start = 5, 70
    out = _torchrun(2, [os.path.join(ROOT, "examples", "whisper", "run.py"), "--engine_dir", str(eng), "--synthetic", str(n),
                        "--synthetic_start", str(start), "--dump_ids", str(tmp_path / "ids.json")])
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    assert "on 2 rank(s)" in out.stdout
    got = json.load(open(tmp_path / "ids.json"))
    enc = wt.WhisperEncoderEngine((eng / "WhisperEncoder.engine").read_bytes())
    dec = wt.WhisperDecoderEngine((eng / "WhisperDecoder.engine").read_bytes(), cfg)
    assert len(got) == n
    for i in range(n):      # per utterance: a sharded batch pads to ITS batch's longest row, so compare up to the row's own end
        one = dec.generate(enc(torch.from_numpy(wt.synthetic.make_mel(cfg, index=start + i, batch=1)).cuda())).cpu().tolist()[0]
        assert got[i][:len(one)] == one and all(t == cfg["pad_token_id"] for t in got[i][len(one):]), (i, got[i], one)


def test_cal_wer_under_torchrun_two_ranks(wt, tmp_path):
    """examples/whisper/cal_wer.py with two ranks on one GPU gives the WER of the single-process run (hypotheses gathered in rank order)."""
    import json
    import subprocess
    z, cfg, weights, mel = load_case("toy-short-eos1_b3")
    eng, ckpt = _write_engine_dir(wt, tmp_path, cfg, weights), tmp_path / "ckpt"
    ckpt.mkdir()
    _toy_vocabulary(cfg, ckpt)
    mels = wt.synthetic.make_mel(cfg, index=70, batch=5)
    refs = ["alpha beta", "gamma", "delta epsilon zeta", "eta", "theta iota"]
    pickle.dump([(mels[i], refs[i]) for i in range(5)], open(tmp_path / "librispeech.cache", "wb"))
    script = [os.path.join(ROOT, "examples", "whisper", "cal_wer.py"), "--whisper", str(ckpt), "--engine_dir", str(eng), "--cache",
              str(tmp_path / "librispeech.cache"), "--batch", "2"]
    single = subprocess.run([sys.executable] + script, capture_output=True, text=True, timeout=300)
    assert single.returncode == 0, single.stderr[-2000:]
    multi = _torchrun(2, script)
    assert multi.returncode == 0, (multi.stdout[-1500:], multi.stderr[-3000:])
    wer = lambda text: [ln for ln in text.splitlines() if ln.startswith("WER:")][-1].split()[1]
    assert wer(single.stdout) == wer(multi.stdout) and "2 rank(s)" in multi.stdout


def test_run_py_compare_ignores_batch_padding(wt, tmp_path):
    """--compare: fast-path rows come from batches padded to the longest row, Session rows stop at their own EOS (batch 1, like the
    reference's own compare): rows must be trimmed behind their first EOS before the comparison."""
    import subprocess
    z, cfg, weights, mel = load_case("toy-short-eos1_b3")
    eng = _write_engine_dir(wt, tmp_path, cfg, weights)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "whisper", "run.py"), "--engine_dir", str(eng), "--synthetic", "3",
                          "--synthetic_start", str(int(z["mel_index"])), "--compare", "--dump_ids", str(tmp_path / "ids.json")],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "Compare Result: same [3], diff [0]" in out.stdout, out.stdout[-1500:]
    import json
    got = np.array(json.load(open(tmp_path / "ids.json")))
    np.testing.assert_array_equal(got, z["ids"])          # and the fast path reproduces the reference-recorded golden (early-EOS row padded)


def test_unfused_argmax_path_still_matches_golden(wt, tmp_path):
    """WT_NO_FUSED_ARGMAX=1 (A/B switch, read once per process): logits to HBM + greedy_select_kernel + greedy_finish_kernel.
    Run in a fresh process through examples/whisper/run.py; ids must equal the reference-recorded golden (early-EOS row padded)."""
    import json
    import subprocess
    z, cfg, weights, mel = load_case("toy-short-eos1_b3")
    eng = _write_engine_dir(wt, tmp_path, cfg, weights)
    env = dict(os.environ, WT_NO_FUSED_ARGMAX="1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "whisper", "run.py"), "--engine_dir", str(eng), "--synthetic", "3",
                          "--synthetic_start", str(int(z["mel_index"])), "--dump_ids", str(tmp_path / "ids.json")],
                         capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    np.testing.assert_array_equal(np.array(json.load(open(tmp_path / "ids.json"))), z["ids"])
