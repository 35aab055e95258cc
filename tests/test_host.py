"""CPU: host-side logic — weight-pack format, builder/parameter-tree surface, logits processors, greedy loop,
and that the C-ABI library loads and exports every symbol its headers declare (no compute without a GPU)."""
import ctypes
import os
import re
import sys

import numpy as np
import pytest
import torch

import cpu_ref
import whisper_trtllm_amd as wt
from whisper_trtllm_amd import _lib, convert, engine_pack, generation, synthetic

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    for header in ("whisper_trtllm_amd.h", "whisper_trtllm_amd_debug.h"):
        text = open(os.path.join(ROOT, "include", header)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names = set(re.findall(r"\b(wt_[a-z_0-9]+)\s*\(", text))
        assert names, header
        for n in sorted(names):
            assert hasattr(lib, n), f"{n} declared in {header} but not exported"
    assert lib.wt_abi_version() == 3


def test_open_rejects_garbage_and_reports():
    lib = _lib.load()
    h = ctypes.c_void_p()
    assert lib.wt_engine_open(b"x" * 10, 10, 0, ctypes.byref(h)) != 0 and "too small" in _lib.last_error()
    blob = bytearray(convert.build_encoder_engine(synthetic.get_config("toy-short"), synthetic.make_weights(synthetic.get_config("toy-short"), 1)))
    bad = bytes(b"NOTMAGIC" + blob[8:])
    assert lib.wt_engine_open(bad, len(bad), 0, ctypes.byref(h)) != 0 and "magic" in _lib.last_error()
    trunc = bytes(blob[:len(blob) // 2])
    assert lib.wt_engine_open(trunc, len(trunc), 0, ctypes.byref(h)) != 0 and "truncated" in _lib.last_error()
    assert not h.value


def test_open_rejects_wrapping_offsets_through_the_c_abi():
    """64-bit table/tensor offsets chosen so that `offset + size` wraps past 2^64: the C-ABI must refuse them before any
    device work (the old checks were sums of file-controlled terms).  The exhaustive version of this runs under
    ASan/UBSan in tests/test_host_sanitizer.py; here the shipped library itself is checked."""
    import struct
    lib = _lib.load()
    cfg = synthetic.get_config("toy-short")
    good = bytes(convert.build_encoder_engine(cfg, synthetic.make_weights(cfg, 1)))
    n_tensors, = struct.unpack_from("<I", good, 20)
    table_off, data_off, total = struct.unpack_from("<QQQ", good, 120)
    assert total == len(good)

    def patched(off, fmt, val):
        b = bytearray(good)
        struct.pack_into(fmt, b, off, val)
        return bytes(b)
    cases = {
        "table_off wraps": patched(120, "<Q", 2 ** 64 - n_tensors * 152),
        "table_off near 2^64": patched(120, "<Q", 2 ** 64 - 100),
        "n_tensors huge": patched(20, "<I", 2 ** 32 - 1),
        "data_off inside table": patched(128, "<Q", table_off + 16),
        "tensor offset wraps": patched(table_off + 136, "<Q", 2 ** 64 - 4096),
        "tensor nbytes huge": patched(table_off + 144, "<Q", 2 ** 64 - 1),
        "dimension product wraps": patched(table_off + 104, "<q", 2 ** 62),
    }
    h = ctypes.c_void_p()
    for what, blob in cases.items():
        rc = lib.wt_engine_open(blob, len(blob), 0, ctypes.byref(h))
        assert rc == -22 and not h.value, (what, rc, _lib.last_error())
    half = bytes(convert.build_encoder_engine(cfg, synthetic.make_weights(cfg, 1), precision="float16"))
    bad = bytearray(half)
    struct.pack_into("<i", bad, 24 + 4 * 3, cfg["d_model"] // 2)          # cfg[CFG_FFN] < d_model in an fp16 engine
    assert lib.wt_engine_open(bytes(bad), len(bad), 0, ctypes.byref(h)) == -22 and "ffn_dim >= d_model" in _lib.last_error()


def test_product_fails_loudly_without_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    cfg = synthetic.get_config("toy-short")
    blob = convert.build_encoder_engine(cfg, synthetic.make_weights(cfg, 1))
    with pytest.raises(RuntimeError):
        wt.Session.from_serialized_engine(blob)   # no CPU fallback anywhere in the product path


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under the package (or the example scripts) may import it."""
    for top in ("whisper-trtllm_amd", "examples"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, top)):
            for f in files:
                if f.endswith((".py", ".hip", ".h")):
                    src = open(os.path.join(dirpath, f)).read()
                    assert "cpu_ref" not in src, f
                    assert not re.search(r"^\s*(from|import)\s+oracle", src, re.M), f


def test_engine_pack_roundtrip_and_layout():
    cfg = synthetic.get_config("toy-wide")
    W = synthetic.make_weights(cfg, 4)
    info, t = engine_pack.unpack(convert.build_encoder_engine(cfg, W))
    d, C = cfg["d_model"], cfg["num_mel_bins"]
    assert info["kind"] == engine_pack.KIND_ENCODER and info["d_model"] == d and info["n_heads"] == cfg["encoder_attention_heads"]
    # conv weights are k-major implicit-GEMM rows: w[co][k*C + ci] == conv.weight[co][ci][k]
    c1 = W["model.encoder.conv1.weight"]
    assert t["conv1.weight"].shape == (d, 3 * C)
    np.testing.assert_array_equal(t["conv1.weight"][5, 2 * C + 7], c1[5, 7, 2])
    np.testing.assert_array_equal(t["conv2.weight"][3, 1 * d + 9], W["model.encoder.conv2.weight"][3, 9, 1])
    qkv_b = t["layers.1.self_attn.qkv.bias"]
    np.testing.assert_array_equal(qkv_b[d:2 * d], np.zeros(d, np.float32))       # k has no bias (build_encoder.py:79)
    np.testing.assert_array_equal(t["layers.1.self_attn.qkv.weight"][d:2 * d], W["model.encoder.layers.1.self_attn.k_proj.weight"])
    info, t = engine_pack.unpack(convert.build_decoder_engine(cfg, W))
    assert info["kind"] == engine_pack.KIND_DECODER and info["tied_proj_out"] == 1 and "proj_out.weight" not in t
    np.testing.assert_array_equal(t["layers.0.encoder_attn.kv.weight"][d:], W["model.decoder.layers.0.encoder_attn.v_proj.weight"])
    np.testing.assert_array_equal(t["layers.0.encoder_attn.kv.bias"][:d], np.zeros(d, np.float32))
    # an untied projection is stored separately
    W2 = dict(W)
    W2["proj_out.weight"] = W["proj_out.weight"] + 1.0
    info, t = engine_pack.unpack(convert.build_decoder_engine(cfg, W2))
    assert info["tied_proj_out"] == 0 and "proj_out.weight" in t


def test_parameter_tree_surface():
    """Attribute paths and shape-checked `.value` assignment used by build_encoder.py:71-91 / build_decoder.py:71-101."""
    enc = wt.models.WhisperEncoder(d_model=128, encoder_layers=2, encoder_attention_heads=2, encoder_ffn_dim=256)
    assert enc.conv1.weight.value.shape == (128, 80, 1, 3) and enc.conv2.weight.value.shape == (128, 128, 1, 3)
    assert enc.embed_positions_weight.shape == (1, 1500, 128)
    assert enc.layers[1].self_attn.qkv.weight.value.shape == (384, 128)
    with pytest.raises(AssertionError):
        enc.layers[0].fc1.weight.value = np.zeros((3, 3), np.float32)
    names = dict(enc.named_parameters())
    assert "layers.0.self_attn.qkv.weight" in names and "layer_norm.bias" in names and "conv1.bias" in names
    dec = wt.models.WhisperDecoder(d_model=128, decoder_layers=2, decoder_attention_heads=2, decoder_ffn_dim=256, vocab_size=512)
    names = dict(dec.named_parameters())
    assert "layers.1.encoder_attn.k_proj.weight" in names and "layers.1.encoder_attn.k_proj.bias" not in names
    assert "proj_out.weight" in names and "proj_out.bias" not in names
    ins = dec.prepare_inputs()
    assert [t.name for t in ins[1:]] == ["encoder_hidden_states", "self_past_key", "self_past_value", "cross_past_key",
                                         "cross_past_value", "past_self_cache_mask", "past_cross_cache_mask"]
    assert ins[0].data.name == "data" and ins[0].data.dtype is wt.trt.int32 and ins[2].shape == (2, 2, -1, 64)
    with pytest.raises(RuntimeError):
        enc(enc.prepare_inputs())               # must be traced inside net_guard
    b = wt.Builder()
    with pytest.raises(ValueError):
        b.create_builder_config(precision="int4")
    net = b.create_network()
    assert b.build_engine(net, b.create_builder_config(precision="float32")) is None   # nothing traced -> None like the reference


def test_logits_processors_match_oracle_rules():
    cfg = synthetic.get_config("toy-short")
    procs = generation.get_logits_processor(cfg, 1)
    g = torch.Generator().manual_seed(0)
    for cur_len in (1, 2, 3, 7):
        ids = torch.zeros(3, cur_len, dtype=torch.long)
        scores = torch.randn(3, cfg["vocab_size"], generator=g)
        want = cpu_ref.apply_logits_processors(cfg, cur_len, 1, scores)
        got = procs(ids, scores.clone())
        assert torch.equal(torch.nan_to_num(got, neginf=-1e30), torch.nan_to_num(want, neginf=-1e30))
    assert generation.get_stopping_criteria(cfg)(torch.zeros(1, cfg["max_length"]))
    assert not generation.get_stopping_criteria(cfg)(torch.zeros(1, cfg["max_length"] - 1))


def test_python_greedy_loop_equals_oracle():
    """generation.greedy_search (the run.py:171-227 loop used on the Session path) driven by the oracle's decoder."""
    from conftest import load_case
    z, cfg, weights, mel = load_case("toy-short-eos1_b3")
    W = cpu_ref.to_torch(weights)
    with torch.no_grad():
        enc = cpu_ref.encoder_forward(W, cfg, torch.from_numpy(mel))
        model = lambda ids, e, past: cpu_ref.decoder_forward(W, cfg, ids.long(), e, past)
        start = torch.full((3, 1), cfg["decoder_start_token_id"], dtype=torch.long)
        out = generation.greedy_search(model, enc, start, generation.get_logits_processor(cfg, 1),
                                       generation.get_stopping_criteria(cfg), cfg["pad_token_id"], cfg["eos_token_id"])
    np.testing.assert_array_equal(out.numpy(), z["ids"])


def test_synthetic_weights_are_reproducible_and_complete():
    cfg = synthetic.get_config("whisper-tiny.en")
    names = [n for n, _, _ in synthetic.weight_specs(cfg)]
    assert len(names) == len(set(names)) and len(names) == 4 * 15 + 4 * 24 + 7 + 4
    a = synthetic._tensor("model.encoder.conv1.weight", (384, 80, 3), "conv", 7)
    b = synthetic._tensor("model.encoder.conv1.weight", (384, 80, 3), "conv", 7)
    assert a.dtype == np.float32 and np.array_equal(a, b)
    assert not np.array_equal(a, synthetic._tensor("model.encoder.conv1.weight", (384, 80, 3), "conv", 8))
    assert len(cfg["suppress_tokens"]) == 90 and cfg["vocab_size"] == 51864


def test_utterance_sharding():
    from whisper_trtllm_amd import sharding
    for total in (0, 1, 7, 64, 73):
        for world in (1, 2, 3, 8):
            spans = [sharding.utterance_shard(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [e - b for b, e in spans]
            assert max(sizes) - min(sizes) <= 1
    assert sharding.batches(3, 20, 8) == [(3, 11), (11, 19), (19, 20)]
    with pytest.raises(ValueError):
        sharding.utterance_shard(8, 2, 2)


@pytest.mark.parametrize("fmt", ["safetensors", "bin"])
def test_build_scripts_from_a_local_hf_checkpoint_dir(tmp_path, fmt):
    """`build_encoder.py` / `build_decoder.py --whisper <dir>` on a local HF-format checkpoint (config.json,
    generation_config.json, model.safetensors or pytorch_model.bin with HF tensor names; `proj_out.weight` absent because it is tied).
    The directory is written by this test from the seeded toy weights; the engines must carry exactly those tensors."""
    import json
    import pickle
    import subprocess
    import torch
    import whisper_trtllm_amd as w
    cfg = w.synthetic.get_config("toy")
    weights = w.synthetic.make_weights(cfg, 5)
    ckpt = tmp_path / "whisper-toy.en"
    ckpt.mkdir()
    gen_keys = ("suppress_tokens", "begin_suppress_tokens", "forced_decoder_ids", "max_length", "decoder_start_token_id", "eos_token_id", "pad_token_id")
    json.dump({k: v for k, v in cfg.items() if k not in gen_keys}, open(ckpt / "config.json", "w"))
    json.dump({k: cfg[k] for k in gen_keys if k in cfg}, open(ckpt / "generation_config.json", "w"))
    sd = {k: np.ascontiguousarray(v) for k, v in weights.items() if k != "proj_out.weight"}
    if fmt == "safetensors":
        from safetensors.numpy import save_file
        save_file(sd, str(ckpt / "model.safetensors"))
    else:
        torch.save({k: torch.from_numpy(v) for k, v in sd.items()}, str(ckpt / "pytorch_model.bin"))
    eng = tmp_path / "eng"
    env = dict(os.environ, PYTHONPATH=ROOT)
    for script in ("build_encoder.py", "build_decoder.py"):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "whisper", script), "--whisper", str(ckpt), "--engine_dir", str(eng)],
                             capture_output=True, text=True, env=env, timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
    config = pickle.loads((eng / "config.pkl").read_bytes())
    for k in gen_keys:
        assert config[k] == cfg[k]
    info, t = w.engine_pack.unpack((eng / "WhisperEncoder.engine").read_bytes())
    assert info["kind"] == w.engine_pack.KIND_ENCODER and info["d_model"] == cfg["d_model"]
    np.testing.assert_array_equal(t["layers.0.fc1.weight"], weights["model.encoder.layers.0.fc1.weight"])
    info, t = w.engine_pack.unpack((eng / "WhisperDecoder.engine").read_bytes())
    assert info["kind"] == w.engine_pack.KIND_DECODER and info["tied_proj_out"] == 1 and "proj_out.weight" not in t
    np.testing.assert_array_equal(t["embed_tokens.weight"], weights["model.decoder.embed_tokens.weight"])
    np.testing.assert_array_equal(t["layers.1.encoder_attn.kv.weight"][:cfg["d_model"]], weights["model.decoder.layers.1.encoder_attn.k_proj.weight"])
    # and the blob equals the one built in-process from the same weights
    assert (eng / "WhisperDecoder.engine").read_bytes() == bytes(w.convert.build_decoder_engine(cfg, weights))


def test_hot_kernels_do_not_spill(tmp_path):
    """The decode GEMV / attention kernels sit at the 256-VGPR edge by design (activation rows + weight rows in registers).  A spill
    is a scratch access = a vector-memory operation that queues behind every weight load in flight: round 2 lost 0.6-1 us per GEMV
    launch (10 us on the vocabulary GEMV) to 8-20 bytes of scratch before this check existed.  hipcc cross-compiles without a GPU."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    seen = 0
    for fname, hot in (("kernels_decoder.hip", ("skinny", "dec_attn")),
                       ("kernels_encoder.hip", ("gemm_f32", "enc_attn", "layernorm")),
                       ("kernels_encoder_f16.hip", ("gemm_f16_dma", "enc_attn_f16", "layernorm_h"))):
        src = os.path.join(ROOT, "whisper-trtllm_amd", "csrc", fname)
        r = subprocess.run([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-c", src, "-o", str(tmp_path / "k.o"),
                            "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        name = None
        for line in r.stderr.splitlines():
            m = re.search(r"Function Name: (\S+)", line)
            if m:
                name = m.group(1)
            m = re.search(r"ScratchSize \[bytes/lane\]: (\d+)", line)
            if m and name and any(h in name for h in hot):
                seen += 1
                assert int(m.group(1)) == 0, f"{name} spills {m.group(1)} bytes per lane"
    assert seen >= 30, seen


def test_pipeline_worker_dispatch_host_logic():
    """runtime.run_workers (the host side of WhisperPipeline.transcribe, no GPU): results in input order whatever worker produced
    them, items handed out dynamically (a slow item does not hold the others back), setup / teardown once per thread, fewer items than
    workers, and the first exception re-raised in the caller after every worker has returned."""
    import threading
    import time
    from whisper_trtllm_amd.runtime import run_workers
    seen, ups, downs = [], [], []
    lock = threading.Lock()

    def item(k, i):
        time.sleep(0.05 if i == 0 else 0.001)       # item 0 is slow: its worker takes fewer items
        with lock:
            seen.append((k, i))
        return i * i

    out = run_workers(40, 3, item, setup=lambda k: ups.append(k), teardown=lambda k: downs.append(k))
    assert out == [i * i for i in range(40)]
    assert sorted(i for _, i in seen) == list(range(40)) and sorted(ups) == [0, 1, 2] and sorted(downs) == [0, 1, 2]
    per_worker = {k: sum(1 for kk, _ in seen if kk == k) for k in range(3)}
    slow = next(k for k, i in seen if i == 0)
    assert per_worker[slow] < max(per_worker.values())                    # dynamic hand-out, not round-robin
    assert run_workers(0, 4, item) == [] and run_workers(2, 8, item) == [0, 1]

    def bad(k, i):
        if i == 5:
            raise KeyError("item 5")
        return i
    with pytest.raises(KeyError):
        run_workers(50, 4, bad)


def test_under_rocprof_detection(monkeypatch):
    """runtime.under_rocprof(): the guard WhisperPipeline uses to clamp itself to one worker under rocprofv3."""
    import whisper_trtllm_amd as w
    for k in list(__import__("os").environ):
        if k.startswith("ROCPROF_") or k in ("ROCP_TOOL_LIBRARIES", "LD_PRELOAD"):
            monkeypatch.delenv(k, raising=False)
    assert not w.runtime.under_rocprof()
    monkeypatch.setenv("LD_PRELOAD", "/opt/rocm/lib/rocprofiler-sdk/librocprofiler-sdk-tool.so")
    assert w.runtime.under_rocprof()
    monkeypatch.delenv("LD_PRELOAD")
    monkeypatch.setenv("ROCPROF_OUTPUT_PATH", "/tmp/x")
    assert w.runtime.under_rocprof()


def test_continuous_scheduler_on_a_simulated_stream():
    """runtime.transcribe_continuous (host scheduler of the continuous mode) against a SIMULATED DecodeStream (no GPU): every utterance
    comes back exactly once in input order whatever the slot count, chunk size and cache-pool size; the pool is never over-committed;
    the queue is topped up while utterances remain (no slot runs dry before the input is exhausted)."""
    import numpy as np
    import torch
    import whisper_trtllm_amd as w

    class FakeStream:
        def __init__(self, slots, pool_rows, lengths):
            self.slots, self.pool_rows, self.lengths = slots, pool_rows, lengths
            self._open, self.n_steps, self.n_submitted = [], 0, 0
            self.rows = {}            # handle -> [utterance id, remaining steps, state]  state: 0 waiting, 1 decoding, 2 finished
            self.queue, self.active, self.dry_steps, self.exhausted = [], [], 0, False

        def free_rows(self):
            return self.pool_rows - len(self._open)

        def submit(self, hidden, force_eos_steps=None):
            n = hidden.shape[0]
            assert 1 <= n <= 16 and n <= self.free_rows(), "pool over-committed"
            used = set(self._open)
            hs = [r for r in range(self.pool_rows) if r not in used][:n]
            for j, h in enumerate(hs):
                self.rows[h] = [int(hidden[j, 0, 0]), None, 0]
                self.queue.append(h)
            self._open.extend(hs)
            self.n_submitted += n
            return hs

        def _admit(self):
            while len(self.active) < self.slots and self.queue:
                h = self.queue.pop(0)
                self.rows[h][1], self.rows[h][2] = self.lengths[self.rows[h][0]], 1
                self.active.append(h)

        def run(self, min_waiting=0, lookahead=0):
            self._admit()
            while True:
                if all(self.rows[h][2] == 2 for h in self._open):
                    break
                if min_waiting > 0 and len(self.queue) < min_waiting:
                    break
                self.n_steps += 1
                if len(self.active) < self.slots and not self.exhausted:
                    self.dry_steps += 1
                for h in list(self.active):
                    self.rows[h][1] -= 1
                    if self.rows[h][1] == 0:
                        self.rows[h][2] = 2
                        self.active.remove(h)
                self._admit()
            return sum(1 for h in self._open if self.rows[h][2] == 2), len(self.queue)

        def collect(self):
            out = [(h, np.full(self.lengths[self.rows[h][0]] + 1, self.rows[h][0], dtype=np.int32)) for h in self._open if self.rows[h][2] == 2]
            self._open = [h for h in self._open if self.rows[h][2] != 2]
            return out

    class FakeDec:
        def __init__(self, lengths):
            self.lengths, self.last = lengths, None

        def stream(self, slots=8, pool_rows=0, max_length=None):
            self.last = FakeStream(slots, pool_rows or 4 * slots, self.lengths)
            return self.last

    rng = np.random.default_rng(5)
    for n, slots, chunk, pool in [(37, 8, 8, 0), (5, 8, 8, 0), (64, 3, 16, 0), (23, 2, 2, 3), (1, 1, 1, 2), (40, 4, 16, 5), (0, 4, 4, 0)]:
        lengths = [int(x) for x in rng.integers(1, 40, size=max(n, 1))]
        mels = torch.arange(n, dtype=torch.float32).reshape(n, 1, 1).repeat(1, 2, 3) if n else torch.zeros(0, 2, 3)
        dec = FakeDec(lengths)
        seen = []

        def enc(x, dec=dec, seen=seen):     # the "encoder": utterance id in element [j, 0, 0]; the input is exhausted once the last id went by
            seen.extend(int(v) for v in x[:, 0, 0])
            if dec.last is not None and len(seen) == n:
                dec.last.exhausted = True
            return x

        stats = {}
        got = w.runtime.transcribe_continuous(enc, dec, mels, slots=slots, chunk=chunk, pool_rows=pool, stats=stats)
        assert len(got) == n and seen == list(range(n))
        for i in range(n):
            assert len(got[i]) == lengths[i] + 1 and (got[i] == i).all()
        if n:
            assert stats["row_steps"] == sum(lengths[:n]) and stats["steps"] == dec.last.n_steps
            if pool == 0 and chunk >= slots:     # a roomy pool and chunks of at least `slots`: a slot is idle only once the input is exhausted
                assert dec.last.dry_steps == 0, (n, slots, chunk, dec.last.dry_steps)
