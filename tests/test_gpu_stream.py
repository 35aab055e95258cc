"""GPU: continuous decoding (wt_decoder_stream_*, runtime.DecodeStream / transcribe_continuous).

The reference transcribes one clip at a time, so every utterance stops at its own EOS (TL/examples/whisper/run.py:219-226; dataset loop
cal_wer.py:249-287).  The continuous mode keeps `slots` rows decoding and refills a slot ON THE DEVICE the moment its utterance stops.
What must hold, whatever slot an utterance lands in, whenever it is admitted and whoever its neighbours are:
  * its ids equal the ORACLE's greedy search of that utterance alone (oracle/cpu_ref.py: the bundled HF path restated), and
  * they equal the engine's own batch-1 decode of it (row independence of every kernel of the step);
  * every utterance comes back exactly once, in the order asked for; runs are bitwise reproducible."""
import ctypes
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def wt():
    import whisper_trtllm_amd as w
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU is visible")
    w._lib.load()
    return w


def _engines(wt, cfg, weights, precision="float32"):
    enc = wt.WhisperEncoderEngine(wt.convert.build_encoder_engine(cfg, weights))
    dec = wt.WhisperDecoderEngine(wt.convert.build_decoder_engine(cfg, weights, precision=precision), cfg)
    return enc, dec


def _oracle_rows(cfg, weights, mel, eos_steps):
    import cpu_ref
    W = cpu_ref.to_torch(weights)
    out = []
    with torch.no_grad():
        h = cpu_ref.encoder_forward(W, cfg, torch.from_numpy(mel))
        for i in range(mel.shape[0]):
            ids = cpu_ref.greedy_search(W, cfg, h[i:i + 1], force_eos_at=None if eos_steps is None or eos_steps[i] < 0 else [eos_steps[i]])
            out.append(ids[0].numpy().astype(np.int32))
    return out


@pytest.mark.parametrize("slots,chunk", [(3, 4), (8, 8), (1, 2), (5, 16)])
def test_ragged_arrivals_decode_like_single_utterances(wt, slots, chunk):
    """21 utterances with transcript lengths between 1 and 19 tokens (forced per utterance, the bench's variable-length device), three of
    them running to max_length: every id row equals the oracle's batch-1 greedy search of that utterance, in input order."""
    cfg = wt.synthetic.get_config("toy-short")
    cfg["max_length"] = 24
    weights = wt.synthetic.make_weights(cfg, 12)
    enc, dec = _engines(wt, cfg, weights)
    n = 21
    eos = [(5 * i + 1) % 19 for i in range(n)]
    for i in (4, 11, 17):
        eos[i] = -1                                   # never forced: these run to max_length unless the model says EOS itself
    mel = wt.synthetic.make_mel(cfg, index=700, batch=n)
    want = _oracle_rows(cfg, weights, mel, eos)
    got = wt.transcribe_continuous(enc, dec, torch.from_numpy(mel).cuda(), slots=slots, chunk=chunk, force_eos_steps=eos)
    assert len(got) == n
    for i in range(n):
        np.testing.assert_array_equal(got[i], want[i], err_msg=f"utterance {i} (eos step {eos[i]})")
        if eos[i] >= 0:
            assert len(got[i]) == eos[i] + 2 and got[i][-1] == cfg["eos_token_id"]
        else:
            assert len(got[i]) <= cfg["max_length"]
    assert any(len(g) == cfg["max_length"] for g in got)          # the max_length stop of a single row was exercised
    # and the engine's own batch-1 decode of a few of them (another cross-attention split plan than 8 slots use)
    for i in (0, 4, 13, 20):
        one = dec.generate(enc(torch.from_numpy(mel[i:i + 1]).cuda()), force_eos_steps=[eos[i]]).cpu().numpy()[0]
        np.testing.assert_array_equal(got[i], one[:len(got[i])])
    # bitwise reproducible, and the aligned path on the same engine is undisturbed by the stream that ran before it
    again = wt.transcribe_continuous(enc, dec, torch.from_numpy(mel).cuda(), slots=slots, chunk=chunk, force_eos_steps=eos)
    for a, b in zip(got, again):
        np.testing.assert_array_equal(a, b)


def test_natural_eos_and_golden_after_a_stream(wt):
    """A model that emits EOS by itself (the reference-recorded golden `toy-short-eos1_b3`: eos among its forced / likely tokens): the
    continuous mode stops each utterance where the reference's generate() did, and the aligned batch path still reproduces the golden
    on the same engine afterwards (the two modes share workspace, tables and graphs)."""
    from conftest import load_case
    z, cfg, weights, mel = load_case("toy-short-eos1_b3")
    enc, dec = _engines(wt, cfg, weights)
    x = torch.from_numpy(mel).cuda()
    ids_ref = z["ids"]
    got = wt.transcribe_continuous(enc, dec, x, slots=2, chunk=2)
    eos, pad = cfg["eos_token_id"], cfg["pad_token_id"]
    for b in range(mel.shape[0]):
        row = ids_ref[b]
        stop = np.nonzero(row[1:] == eos)[0]
        n = int(stop[0]) + 2 if len(stop) else len(row)
        np.testing.assert_array_equal(got[b], row[:n].astype(np.int32))
    batch = dec.generate(enc(x)).cpu().numpy()
    np.testing.assert_array_equal(batch, ids_ref[:, :batch.shape[1]])


def test_tiny_en_slots8_matches_batch1_engine_and_oracle(wt):
    """whisper-tiny.en sizes, 8 slots, 20 utterances with LibriSpeech-like lengths: ids equal the engine's batch-1 decode for every
    utterance and the oracle's for three of them (the oracle's margins on this seed are healthy: asserted)."""
    import cpu_ref
    cfg = wt.synthetic.get_config("whisper-tiny.en")
    cfg["max_length"] = 64
    weights = wt.synthetic.make_weights(cfg, 77)
    enc, dec = _engines(wt, cfg, weights)
    n = 20
    _dur, eos = wt.synthetic.librispeech_like_lengths(n, seed=3, max_length=64)
    eos = [int(e) for e in eos]
    mel_np = wt.synthetic.make_mel(cfg, index=300, batch=n)
    mel = torch.from_numpy(mel_np).cuda()
    stats = {}
    got = wt.transcribe_continuous(enc, dec, mel, slots=8, chunk=8, force_eos_steps=eos, stats=stats)
    assert stats["row_steps"] == sum(len(g) - 1 for g in got)
    for i in range(n):
        one = dec.generate(enc(mel[i:i + 1]), force_eos_steps=[eos[i]]).cpu().numpy()[0]
        np.testing.assert_array_equal(got[i], one[:len(got[i])], err_msg=f"utterance {i}")
        assert got[i][-1] == cfg["eos_token_id"] or len(got[i]) == 64
    W = cpu_ref.to_torch(weights)
    torch.set_num_threads(16)
    with torch.no_grad():
        for i in (0, 7, 19):
            h = cpu_ref.encoder_forward(W, cfg, torch.from_numpy(mel_np[i:i + 1]))
            ids, logits = cpu_ref.greedy_search(W, cfg, h, force_eos_at=[eos[i]], return_logits=True)
            top2 = torch.topk(logits[:, 1:], 2, dim=-1).values
            assert (top2[..., 0] - top2[..., 1]).min().item() > 1e-3, "oracle margin too thin on this seed"
            np.testing.assert_array_equal(got[i], ids[0].numpy().astype(np.int32))


def test_fp16_decoder_engine_streams_too(wt):
    """fp16 decoder engines (half weights, half resident caches incl. the cache POOL): continuous ids equal the same engine's batch-1 ids."""
    cfg = wt.synthetic.get_config("toy-short")
    cfg["max_length"] = 20
    weights = wt.synthetic.make_weights(cfg, 55)
    enc, dec = _engines(wt, cfg, weights, precision="float16")
    n = 9
    eos = [3 + (4 * i) % 13 for i in range(n)]
    mel = torch.from_numpy(wt.synthetic.make_mel(cfg, index=500, batch=n)).cuda()
    got = wt.transcribe_continuous(enc, dec, mel, slots=4, chunk=3, force_eos_steps=eos)
    for i in range(n):
        one = dec.generate(enc(mel[i:i + 1]), force_eos_steps=[eos[i]]).cpu().numpy()[0]
        np.testing.assert_array_equal(got[i], one[:len(got[i])])


def test_small_pool_and_c_abi_errors(wt):
    """A cache pool of slots + 1 rows (the scheduler must wait for rows to free up), and the C-ABI's error paths: submit without an
    open stream, more utterances than free rows, a handle that holds nothing, wt_decoder_run on a streaming handle."""
    cfg = wt.synthetic.get_config("toy-short")
    cfg["max_length"] = 16
    weights = wt.synthetic.make_weights(cfg, 12)
    enc, dec = _engines(wt, cfg, weights)
    lib, h = dec.session._lib, dec.session.handle
    mel = torch.from_numpy(wt.synthetic.make_mel(cfg, index=40, batch=10)).cuda()
    hidden = enc(mel)
    handles = (ctypes.c_int32 * 16)()
    s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert lib.wt_decoder_stream_submit(h, hidden.data_ptr(), 2, None, handles, s) == -1          # WT_E_STATE: no stream open
    eos = [2 + (3 * i) % 11 for i in range(10)]
    st = wt.DecodeStream(dec, slots=2, pool_rows=3)
    assert st.free_rows() == 3
    first = st.submit(hidden[:3], eos[:3])
    assert sorted(first) == [0, 1, 2] and st.free_rows() == 0
    assert lib.wt_decoder_stream_submit(h, hidden.data_ptr(), 1, None, handles, s) == -1          # no free cache row
    n = ctypes.c_int()
    assert lib.wt_decoder_stream_collect(h, 7, None, 0, ctypes.byref(n)) == -22                   # handle outside the pool
    cur, nu = ctypes.c_int(), ctypes.c_int()
    assert lib.wt_decoder_run(h, 0, ctypes.byref(cur), ctypes.byref(nu), s) == -1                 # the aligned run loop is refused
    st.run()
    done = dict(st.collect())
    assert sorted(done) == sorted(first) and st.free_rows() == 3
    assert lib.wt_decoder_stream_collect(h, 0, None, 0, ctypes.byref(n)) == -1                    # released: the row holds nothing
    # the scheduler on the same tiny pool: rows are recycled many times over
    got = wt.transcribe_continuous(enc, dec, mel, slots=2, chunk=2, force_eos_steps=eos, pool_rows=3)
    for i in range(10):
        one = dec.generate(enc(mel[i:i + 1]), force_eos_steps=[eos[i]]).cpu().numpy()[0]
        np.testing.assert_array_equal(got[i], one[:len(got[i])])
    for i in range(3):
        np.testing.assert_array_equal(done[first[i]], got[i])


def test_run_py_continuous_batching_and_pipeline_workers(wt, tmp_path):
    """examples/whisper/run.py --batching continuous as a subprocess (2 workers, each with its own continuous stream over blocks of the
    utterances): ids in dataset order, every row ending at its own EOS -- the reference-recorded golden rows cut behind their EOS;
    and WhisperPipeline.transcribe_continuous in-process with 1 / 2 / 3 workers gives identical rows."""
    import json
    import os
    import pickle
    import subprocess
    import sys
    from conftest import load_case
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    z, cfg, weights, mel = load_case("toy-short-eos1_b3")
    eng = tmp_path / "eng"
    eng.mkdir()
    eb, db = wt.convert.build_encoder_engine(cfg, weights), wt.convert.build_decoder_engine(cfg, weights)
    (eng / "WhisperEncoder.engine").write_bytes(eb)
    (eng / "WhisperDecoder.engine").write_bytes(db)
    (eng / "config.pkl").write_bytes(pickle.dumps(cfg))
    out = subprocess.run([sys.executable, os.path.join(root, "examples", "whisper", "run.py"), "--engine_dir", str(eng), "--synthetic", "3",
                          "--synthetic_start", str(int(z["mel_index"])), "--batching", "continuous", "--dump_ids", str(tmp_path / "ids.json")],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    got = json.load(open(tmp_path / "ids.json"))
    eos = cfg["eos_token_id"]
    want = [[int(t) for t in (r[:list(r).index(eos, 1) + 1] if eos in list(r)[1:] else r)] for r in z["ids"]]
    assert got == want
    # the pipeline's own method, more utterances than one block, several worker counts
    n = 23
    mels = torch.from_numpy(wt.synthetic.make_mel(cfg, index=100, batch=n)).cuda()
    fe = [1 + (7 * i) % 15 for i in range(n)]
    rows = {}
    for workers in (1, 2, 3):
        pipe = wt.WhisperPipeline(eb, db, cfg, workers=workers)
        rows[workers] = pipe.transcribe_continuous(mels, slots=4, chunk=4, block=6, force_eos_steps=fe)
        assert len(rows[workers]) == n
        del pipe
    for i in range(n):
        assert rows[1][i][-1] == eos and len(rows[1][i]) <= fe[i] + 2        # its forced EOS, or the model's own before it
        np.testing.assert_array_equal(rows[1][i], rows[2][i])
        np.testing.assert_array_equal(rows[1][i], rows[3][i])


def test_stream_churn_thousands_of_utterances_through_one_open_stream():
    """tools/soak_stream.py at a tenth of its default size: 5000 utterances with random forced lengths through one open stream of 8 slots
    (slots refilled ~5000 times, submissions in chunks of <= 16 while steps are in flight), every id row equal to the batch decode.
    The full run (50,000 utterances: the mailbox's 16-bit step counter and 15-bit admission counter both wrap) is
    profiles/r04d_soak_stream.txt."""
    import subprocess
    import sys
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "soak_stream.py"), "5000"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "stream soak ok: 5000 utterances" in out.stdout
