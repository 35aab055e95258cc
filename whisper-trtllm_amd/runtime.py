"""Runtime boundary: `Session` / `TensorInfo` with the reference's surface (tensorrt_llm/runtime/session.py:28-207),
plus the batched resident-KV fast path (`WhisperEncoderEngine`, `WhisperDecoderEngine`) behind the same C-ABI.
"""
from __future__ import annotations

import contextlib
import ctypes
from dataclasses import dataclass
from typing import Any, Dict, List, Optional, Sequence

from . import _dtypes as trt
from . import _lib
from .logger import logger


def under_rocprof() -> bool:
    """True when rocprofv3's tool library is loaded into this process (it is LD_PRELOADed by the `rocprofv3` launcher and intercepts every
    HIP / HSA call, incl. the AQL packets of a hipGraph replay).  A 4-worker `WhisperPipeline` run under `rocprofv3 --kernel-trace`
    aborted with SIGSEGV in round 3: the faulting frame lies in a module mapped at process start (where the preloaded profiler
    libraries live), below hipGraphLaunch <- enqueue_steps <- wt_decoder_run of a worker thread; the frames could not be symbolised
    (DESIGN.md "The four-worker abort under rocprofv3").  The same runs without the profiler are clean (soak, test suite, driver bench),
    so the hazard is several host threads replaying graphs at once UNDER THE PROFILER -- `WhisperPipeline` therefore runs one worker then."""
    import os
    env = os.environ
    if "rocprofiler" in env.get("LD_PRELOAD", "") or env.get("ROCP_TOOL_LIBRARIES") or any(k.startswith("ROCPROF_") for k in env):
        return True
    try:
        with open("/proc/self/maps") as f:
            return any("rocprofiler-sdk-tool" in line for line in f)
    except OSError:
        return False


@contextlib.contextmanager
def _scoped_stream():
    """Current torch stream handle; synchronised when the scope ends (session.py:14-25)."""
    import torch
    stream = torch.cuda.current_stream()
    try:
        yield stream.cuda_stream
    finally:
        stream.synchronize()


@dataclass
class TensorInfo:
    name: str
    dtype: trt.DataType
    shape: tuple


def _desc(info: TensorInfo) -> _lib.TensorDesc:
    d = _lib.TensorDesc()
    d.name = info.name.encode()
    d.dtype = info.dtype.code
    shape = tuple(int(s) for s in info.shape)
    if len(shape) > _lib.WT_MAX_DIMS:
        raise ValueError(f"{info.name}: rank {len(shape)} too large")
    d.ndim = len(shape)
    for i, s in enumerate(shape):
        d.shape[i] = s
    return d


class Session:
    """`Session.from_serialized_engine(bytes)` -> `infer_shapes` -> `run` (async) like session.py:36-178."""

    def __init__(self, **kwargs):
        self._handle = None
        self._lib = None

    def _init(self, engine_buffer, device: Optional[int] = None):
        import torch
        self._lib = _lib.load()
        if not torch.cuda.is_available():
            raise RuntimeError("whisper-trtllm_amd needs a ROCm GPU (MI355X); there is no CPU execution path")
        dev = torch.cuda.current_device() if device is None else device
        buf = bytes(engine_buffer)
        handle = ctypes.c_void_p()
        _lib.check(self._lib.wt_engine_open(buf, len(buf), dev, ctypes.byref(handle)), "wt_engine_open")
        self._handle = handle
        info = _lib.EngineInfo()
        _lib.check(self._lib.wt_engine_get_info(self._handle, ctypes.byref(info)), "wt_engine_get_info")
        self.info = info
        return self

    @staticmethod
    def from_serialized_engine(engine, device: Optional[int] = None) -> "Session":
        return Session()._init(engine, device)

    def clone(self) -> "Session":
        """A second execution context on the same weights (wt_engine_clone): own workspace / caches / graphs, shared read-only payload."""
        other = Session()
        other._lib = self._lib
        handle = ctypes.c_void_p()
        _lib.check(self._lib.wt_engine_clone(self._handle, ctypes.byref(handle)), "wt_engine_clone")
        other._handle = handle
        other.info = self.info
        return other

    def __del__(self):
        try:
            if self._handle is not None and self._lib is not None:
                self._lib.wt_engine_close(self._handle)
                self._handle = None
        except Exception:
            pass

    @property
    def handle(self):
        return self._handle

    def infer_shapes(self, inputs: List[TensorInfo], context=None) -> Optional[List[TensorInfo]]:
        """Returns the output TensorInfos, or None (after logging) on a name/dtype/shape mismatch — session.py:131-136."""
        n = len(inputs)
        arr = (_lib.TensorDesc * max(n, 1))(*[_desc(i) for i in inputs])
        out = (_lib.TensorDesc * 8)()
        n_out = ctypes.c_int(8)
        rc = self._lib.wt_engine_infer_shapes(self._handle, arr, n, out, ctypes.byref(n_out))
        if rc != 0:
            logger.error(_lib.last_error())
            return None
        return [TensorInfo(out[i].name.decode(), trt.from_code(out[i].dtype), tuple(out[i].shape[k] for k in range(out[i].ndim)))
                for i in range(n_out.value)]

    def run(self, inputs: Dict[str, Any], outputs: Dict[str, Any], stream, context=None) -> bool:
        """Enqueue on `stream` (raw hipStream_t handle as int).  True means enqueued, not finished (session.py:159-160)."""
        def bind(d):
            arr = (_lib.Binding * max(len(d), 1))()
            keep = []
            for i, (name, t) in enumerate(d.items()):
                ptr = t.data_ptr() if hasattr(t, "data_ptr") else int(t)
                nm = name.encode()
                keep.append(nm)
                arr[i].name, arr[i].ptr = nm, ptr
            return arr, keep
        a_in, k1 = bind(inputs)
        a_out, k2 = bind(outputs)
        rc = self._lib.wt_engine_run(self._handle, a_in, len(inputs), a_out, len(outputs), ctypes.c_void_p(stream or 0))
        if rc != 0:
            logger.error(_lib.last_error())
            return False
        return True

    def _debug_run(self, inputs: Dict[str, "torch.Tensor"], context=None) -> Dict[str, "torch.Tensor"]:
        """Synchronous convenience run with freshly allocated outputs (session.py:180-207)."""
        import torch
        infos = [TensorInfo(n, trt.from_torch(t.dtype), tuple(t.shape)) for n, t in inputs.items()]
        outs = self.infer_shapes(infos)
        if outs is None:
            raise RuntimeError(_lib.last_error())
        outputs = {o.name: torch.empty(tuple(o.shape), dtype=trt.torch_dtype(o.dtype), device="cuda") for o in outs}
        with _scoped_stream() as stream:
            if not self.run(inputs, outputs, stream):
                raise RuntimeError(_lib.last_error())
        return outputs


# ----------------------------------------------------------------------------------------------- batched fast path
class WhisperEncoderEngine:
    """mel f32 [B, n_mels, 2*S] on the GPU -> hidden f32 [B, S, d]; asynchronous on the current torch stream."""

    def __init__(self, engine_buffer, device: Optional[int] = None, _session: Optional[Session] = None):
        self.session = _session if _session is not None else Session.from_serialized_engine(engine_buffer, device)
        if self.session.info.kind != 1:
            raise ValueError("not a WhisperEncoder engine")

    def clone(self) -> "WhisperEncoderEngine":
        """Another engine on the same device weights (own workspace): for a second host thread / stream."""
        return WhisperEncoderEngine(None, _session=self.session.clone())

    def __call__(self, mel):
        import torch
        i = self.session.info
        if mel.dtype != torch.float32 or mel.dim() != 3 or mel.shape[1] != i.n_mels or mel.shape[2] != 2 * i.max_source_positions:
            raise ValueError(f"mel must be float32 [B,{i.n_mels},{2 * i.max_source_positions}], got {tuple(mel.shape)} {mel.dtype}")
        if not mel.is_cuda:   # the C-ABI takes DEVICE pointers: a host pointer would fault on the GPU
            raise ValueError("mel must be a CUDA (ROCm) tensor: move it to the GPU first (there is no CPU execution path)")
        mel = mel.contiguous()
        out = torch.empty(mel.shape[0], i.max_source_positions, i.d_model, dtype=torch.float32, device=mel.device)
        stream = torch.cuda.current_stream().cuda_stream
        _lib.check(self.session._lib.wt_encoder_forward(self.session.handle, mel.data_ptr(), mel.shape[0], out.data_ptr(),
                                                         ctypes.c_void_p(stream)), "wt_encoder_forward")
        return out


def _i32_array(values: Sequence[int]):
    vals = [int(v) for v in values]
    return (ctypes.c_int32 * max(len(vals), 1))(*vals), len(vals)


class WhisperDecoderEngine:
    """Greedy decode with the KV cache resident on the device (no per-token host round trip).

    `config` is the HF config dict the reference pickles into engine_dir/config.pkl (run.py:251) — the keys
    read are those of run.py:150-169, 273, 283-284."""

    def __init__(self, engine_buffer, config: dict, device: Optional[int] = None, _session: Optional[Session] = None):
        self.session = _session if _session is not None else Session.from_serialized_engine(engine_buffer, device)
        if self.session.info.kind != 2:
            raise ValueError("not a WhisperDecoder engine")
        self.config = config
        self.max_batch = 16   # utterances per engine call (wt_decoder_begin); larger batches are chunked by generate()

    def clone(self) -> "WhisperDecoderEngine":
        """Another engine on the same device weights (own resident KV caches, step graphs, mailbox): for a second host thread / stream."""
        return WhisperDecoderEngine(None, self.config, _session=self.session.clone())

    def _params(self, max_length, force_eos_step, logits_trace, force_eos_steps=None):
        cfg = self.config
        begin_index = 1 if cfg.get("forced_bos_token_id") is None else 2          # run.py:155-156 with a 1-token prompt
        forced = cfg.get("forced_decoder_ids") or []
        if forced:
            begin_index += forced[-1][0]                                           # run.py:157
        p = _lib.GreedyParams()
        p.decoder_start_token_id = cfg["decoder_start_token_id"]
        p.eos_token_id, p.pad_token_id = cfg["eos_token_id"], cfg["pad_token_id"]
        p.max_length = cfg["max_length"] if max_length is None else max_length
        p.begin_index = begin_index
        self._keep = []
        for field, count, vals in (("suppress_tokens", "n_suppress_tokens", cfg.get("suppress_tokens") or []),
                                   ("begin_suppress_tokens", "n_begin_suppress_tokens", cfg.get("begin_suppress_tokens") or []),
                                   ("forced_decoder_ids", "n_forced", [x for pair in forced for x in pair])):
            arr, n = _i32_array(vals)
            self._keep.append(arr)
            setattr(p, field, ctypes.cast(arr, ctypes.POINTER(ctypes.c_int32)))
            setattr(p, count, n if field != "forced_decoder_ids" else n // 2)
        p.force_eos_step = -1 if force_eos_step is None else force_eos_step
        p.logits_trace = logits_trace.data_ptr() if logits_trace is not None else None
        if force_eos_steps is not None:           # bench-only per-row transcript lengths
            arr, _n = _i32_array(force_eos_steps)
            self._keep.append(arr)
            p.force_eos_steps = ctypes.cast(arr, ctypes.POINTER(ctypes.c_int32))
        return p

    def begin(self, encoder_hidden, max_length=None, force_eos_step=None, logits_trace=None, force_eos_steps=None):
        import torch
        i = self.session.info
        if encoder_hidden.dtype != torch.float32 or tuple(encoder_hidden.shape[1:]) != (i.max_source_positions, i.d_model):
            raise ValueError(f"encoder_hidden must be float32 [B,{i.max_source_positions},{i.d_model}]")
        if not encoder_hidden.is_cuda or (logits_trace is not None and not logits_trace.is_cuda):
            raise ValueError("encoder_hidden / logits_trace must be CUDA (ROCm) tensors: the C-ABI takes device pointers")
        self._enc = encoder_hidden.contiguous()
        self._B = self._enc.shape[0]
        if force_eos_steps is not None and len(force_eos_steps) != self._B:
            raise ValueError(f"force_eos_steps needs one entry per utterance ({self._B}), got {len(force_eos_steps)}")
        self._p = self._params(max_length, force_eos_step, logits_trace, force_eos_steps)
        self._trace = logits_trace
        stream = torch.cuda.current_stream().cuda_stream
        _lib.check(self.session._lib.wt_decoder_begin(self.session.handle, self._enc.data_ptr(), self._B, ctypes.byref(self._p),
                                                       ctypes.c_void_p(stream)), "wt_decoder_begin")

    def steps(self, n: int):
        import torch
        stream = torch.cuda.current_stream().cuda_stream
        _lib.check(self.session._lib.wt_decoder_steps(self.session.handle, n, ctypes.c_void_p(stream)), "wt_decoder_steps")

    def poll(self):
        import torch
        cur, nu, done = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        stream = torch.cuda.current_stream().cuda_stream
        _lib.check(self.session._lib.wt_decoder_poll(self.session.handle, ctypes.byref(cur), ctypes.byref(nu), ctypes.byref(done),
                                                      ctypes.c_void_p(stream)), "wt_decoder_poll")
        return cur.value, nu.value, bool(done.value)

    def run(self, lookahead: int = 0):
        """Drive the decode in flight to its stop test (wt_decoder_run): the host keeps `lookahead` steps queued behind the running
        one and watches a pinned mailbox word instead of synchronising the stream.  Returns (cur_len, n_unfinished)."""
        import torch
        cur, nu = ctypes.c_int(), ctypes.c_int()
        stream = torch.cuda.current_stream().cuda_stream
        _lib.check(self.session._lib.wt_decoder_run(self.session.handle, lookahead, ctypes.byref(cur), ctypes.byref(nu),
                                                     ctypes.c_void_p(stream)), "wt_decoder_run")
        return cur.value, nu.value

    def read_ids(self, cur_len: int):
        import torch
        ml = self._p.max_length
        buf = torch.empty(self._B, ml, dtype=torch.int32, device=self._enc.device)
        stream = torch.cuda.current_stream().cuda_stream
        _lib.check(self.session._lib.wt_decoder_read_ids(self.session.handle, buf.data_ptr(), ml, ctypes.c_void_p(stream)),
                   "wt_decoder_read_ids")
        return buf[:, :cur_len].clone()

    def generate(self, encoder_hidden, max_length=None, force_eos_step=None, logits_trace=None, chunk: int = 0, force_eos_steps=None,
                 lookahead: int = 0):
        """== greedy_search(...) of run.py:171-227 for a batch; returns int32 ids [B, len] on the GPU.

        Default (`chunk` 0): the stop test is followed through the engine's host mailbox (`run`), at most `lookahead` steps are
        enqueued past the stop.  `chunk` > 0 selects the older protocol -- `chunk` steps between two synchronising polls."""
        if encoder_hidden.shape[0] > self.max_batch:
            # larger batches run as consecutive engine batches of <= 16 utterances;
            # rows are independent, so the result is the concatenation (shorter groups are right-padded with pad_token_id)
            import torch
            if logits_trace is not None:
                raise ValueError(f"logits_trace supports at most {self.max_batch} utterances per call")
            parts = [self.generate(encoder_hidden[i:i + self.max_batch], max_length, force_eos_step, None, chunk,
                                   None if force_eos_steps is None else force_eos_steps[i:i + self.max_batch], lookahead)
                     for i in range(0, encoder_hidden.shape[0], self.max_batch)]
            width = max(p.shape[1] for p in parts)
            pad = self.config["pad_token_id"]
            parts = [torch.nn.functional.pad(p, (0, width - p.shape[1]), value=pad) for p in parts]
            return torch.cat(parts, dim=0)
        self.begin(encoder_hidden, max_length, force_eos_step, logits_trace, force_eos_steps)
        if chunk <= 0:
            cur, _nu = self.run(lookahead)
            return self.read_ids(cur)
        cur, done = 1, False
        ml = self._p.max_length
        while not done and cur < ml:
            self.steps(min(chunk, ml - cur))
            cur, _nu, done = self.poll()
        return self.read_ids(cur)

    def stream(self, slots: int = 8, pool_rows: int = 0, max_length=None) -> "DecodeStream":
        """Open a continuous decode on this engine (wt_decoder_stream_*): `slots` rows stay busy, every utterance stops at its own EOS
        and its slot is refilled on the device within the same step.  Ends with the next `begin` / `generate` on this engine."""
        return DecodeStream(self, slots, pool_rows, max_length)

    def set_profiling(self, enabled: bool):
        _lib.check(self.session._lib.wt_engine_set_profiling(self.session.handle, int(enabled)), "wt_engine_set_profiling")

    def time_cross_attention(self, iters: int = 20) -> float:
        """Average launch time (us) of the cross-attention kernel, graph-replayed over the resident caches."""
        import torch
        us = ctypes.c_float()
        stream = torch.cuda.current_stream().cuda_stream
        _lib.check(self.session._lib.wt_decoder_time_cross_attention(self.session.handle, iters, ctypes.byref(us),
                                                                      ctypes.c_void_p(stream)), "wt_decoder_time_cross_attention")
        return float(us.value)

    def time_kernel(self, which: str, iters: int = 20) -> float:
        """Average launch time (us) of one kind of per-layer decode launch ("qkv", "self_attn", "pair", "cross_attn", "cross_out",
        "fc1", "fc2"), graph-replayed over all layers' own weights and caches."""
        import torch
        us = ctypes.c_float()
        stream = torch.cuda.current_stream().cuda_stream
        _lib.check(self.session._lib.wt_decoder_time_kernel(self.session.handle, which.encode(), iters, ctypes.byref(us),
                                                             ctypes.c_void_p(stream)), "wt_decoder_time_kernel")
        return float(us.value)

    def timer(self, which: str):
        t = _lib.KernelTimer()
        _lib.check(self.session._lib.wt_engine_get_timer(self.session.handle, which.encode(), ctypes.byref(t)), "wt_engine_get_timer")
        return t.ms_total, t.launches


class DecodeStream:
    """Continuous greedy decoding (the C-ABI's wt_decoder_stream_*): utterances are SUBMITTED as encoder memory, wait in a device-side
    queue, and take over a decode slot the moment its previous utterance emits EOS (or reaches max_length) -- the per-utterance stop
    the reference gets from transcribing one clip at a time (run.py:219-226), without a batch waiting for its longest row.

        st = dec.stream(slots=8)
        h = st.submit(enc(mel_chunk))             # handles, one per utterance; any number of chunks ahead (pool_rows bounds it)
        st.run(min_waiting=8)                     # steps until fewer than 8 utterances wait (submit more) or everything has finished
        for handle, ids in st.collect(): ...      # finished utterances (numpy int32 rows, start token .. EOS), cache rows released
    """

    def __init__(self, dec: "WhisperDecoderEngine", slots: int, pool_rows: int = 0, max_length=None):
        import torch
        self.dec, self.slots = dec, int(slots)
        self._lib, self._h = dec.session._lib, dec.session.handle
        self._p = dec._params(max_length, None, None)
        self.max_length = int(self._p.max_length)
        self.pool_rows = int(pool_rows) if pool_rows else 4 * self.slots
        stream = torch.cuda.current_stream().cuda_stream
        _lib.check(self._lib.wt_decoder_stream_begin(self._h, self.slots, self.pool_rows, ctypes.byref(self._p), ctypes.c_void_p(stream)),
                   "wt_decoder_stream_begin")
        self._open: List[int] = []          # handles submitted and not collected yet, in submission order
        self._keep: List[Any] = []          # encoder memories whose K/V projection may still be queued on the stream
        self.n_submitted = 0
        self.n_steps = 0

    def free_rows(self) -> int:
        return self.pool_rows - len(self._open)

    def submit(self, encoder_hidden, force_eos_steps: Optional[Sequence[int]] = None) -> List[int]:
        """Queue the utterances of `encoder_hidden` f32 [n, S, d] (n <= 16, n <= free_rows()); returns their handles."""
        import torch
        i = self.dec.session.info
        if encoder_hidden.dtype != torch.float32 or tuple(encoder_hidden.shape[1:]) != (i.max_source_positions, i.d_model) or not encoder_hidden.is_cuda:
            raise ValueError(f"encoder_hidden must be a CUDA float32 [n,{i.max_source_positions},{i.d_model}] tensor")
        h = encoder_hidden.contiguous()
        n = h.shape[0]
        if force_eos_steps is not None and len(force_eos_steps) != n:
            raise ValueError("force_eos_steps needs one entry per utterance")
        fe = None
        if force_eos_steps is not None:
            fe, _ = _i32_array(force_eos_steps)
        handles = (ctypes.c_int32 * n)()
        stream = torch.cuda.current_stream().cuda_stream
        _lib.check(self._lib.wt_decoder_stream_submit(self._h, h.data_ptr(), n, fe, handles, ctypes.c_void_p(stream)), "wt_decoder_stream_submit")
        self._keep.append(h)
        if len(self._keep) > 8:             # projections are stream-ordered: anything this old has long been consumed by later steps
            self._keep.pop(0)
        out = [int(x) for x in handles]
        self._open.extend(out)
        self.n_submitted += n
        return out

    def run(self, min_waiting: int = 0, lookahead: int = 0):
        """Step until everything submitted has finished, or (min_waiting > 0) until fewer than `min_waiting` utterances still wait
        for a slot.  Returns (finished so far, upper bound of utterances still waiting)."""
        import torch
        fin, wait, steps = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        stream = torch.cuda.current_stream().cuda_stream
        _lib.check(self._lib.wt_decoder_stream_run(self._h, int(min_waiting), int(lookahead), ctypes.byref(fin), ctypes.byref(wait),
                                                   ctypes.byref(steps), ctypes.c_void_p(stream)), "wt_decoder_stream_run")
        self.n_steps = steps.value          # decoder steps enqueued since the stream was opened
        return fin.value, wait.value

    def collect(self):
        """[(handle, ids)] of the utterances that have finished since the last call (numpy int32, start token through EOS / max_length);
        their cache rows return to the pool."""
        import numpy as np
        out, still = [], []
        buf = (ctypes.c_int32 * self.max_length)()
        n = ctypes.c_int()
        for h in self._open:
            _lib.check(self._lib.wt_decoder_stream_collect(self._h, h, buf, self.max_length, ctypes.byref(n)), "wt_decoder_stream_collect")
            if n.value > 0:
                out.append((h, np.ctypeslib.as_array(buf)[:n.value].copy()))
            else:
                still.append(h)
        self._open = still
        return out


def transcribe_continuous(enc: "WhisperEncoderEngine", dec: "WhisperDecoderEngine", mels, slots: int = 8, chunk: int = 8,
                          force_eos_steps: Optional[Sequence[int]] = None, max_length=None, stats: Optional[dict] = None,
                          pool_rows: int = 0) -> List[Any]:
    """Encoder + continuous greedy decode of `mels` ([N, n_mels, 2*S] tensor or a list of such rows) in ARRIVAL order: the encoder runs
    on chunks of `chunk` utterances just ahead of the decode, `slots` rows decode at any time.  Returns N numpy id rows (start token
    through EOS) in input order.  This is the host scheduler of the C-ABI's continuous mode -- what cal_wer.py / run.py use for a
    dataset whose transcripts differ in length.  `stats` (optional dict) receives the slot utilisation of the run."""
    import torch
    n = len(mels)
    results: List[Any] = [None] * n
    if n == 0:
        return results
    st = dec.stream(slots=slots, pool_rows=pool_rows, max_length=max_length)
    index_of = {}
    nxt = 0

    def submit_next() -> int:
        nonlocal nxt
        k = min(chunk, n - nxt, st.free_rows())
        if k <= 0:
            return 0
        part = mels[nxt:nxt + k]
        x = part if torch.is_tensor(part) else torch.stack(list(part))
        fe = None if force_eos_steps is None else [int(v) for v in force_eos_steps[nxt:nxt + k]]
        for j, h in enumerate(st.submit(enc(x), fe)):
            index_of[h] = nxt + j
        nxt += k
        return k

    def harvest():
        for h, ids in st.collect():
            results[index_of.pop(h)] = ids

    waiting = 0          # upper bound of the submitted utterances still waiting for a slot
    while True:
        # keep at least `slots` utterances waiting while there are more to come: a freed slot then never runs dry while the next
        # chunk's encoder pass is still ahead
        while nxt < n and waiting < slots and st.free_rows() > 0:
            waiting += submit_next()
        more = nxt < n
        if more and st.free_rows() == 0:     # cache pool exhausted: step until one more utterance has been admitted (= one has finished)
            _, waiting = st.run(min_waiting=max(1, waiting))
        else:
            _, waiting = st.run(min_waiting=slots if more else 0)
        harvest()
        if not more and not st._open:
            break
    assert all(r is not None for r in results)
    if stats is not None:   # slot utilisation = row-steps that produced a token some utterance needed / row-steps paid for
        stats["row_steps"] = int(sum(len(r) - 1 for r in results))
        stats["steps"] = int(st.n_steps)
        stats["slot_utilisation"] = stats["row_steps"] / float(max(1, st.n_steps) * slots)
    return results


class WhisperPipeline:
    """`workers` independent (encoder, decoder) engine pairs on ONE GPU, each driven by its own host thread on its own HIP stream.

    Why: a greedy decode is a chain of ~170 dependent, launch-latency-bound kernels per token -- it leaves most of the chip idle
    most of the time -- while the encoder is MFMA-bound.  Batches are independent (the reference transcribes one clip after the
    other, run.py:262-290), so a second in-flight batch fills the first one's gaps: two workers of batch 8 measure ~1.4x one worker
    on whisper-medium.en (434 vs 312 audio-s/s with 447-step decodes; DESIGN.md section 6 "Workers per GPU").  Every worker is a
    complete execution context (own workspace, resident KV cache, step graphs: ~4 GB for medium.en fp32) on ONE shared, read-only copy
    of the weights (`wt_engine_clone`), so nothing mutable is shared and no lock is taken on the device path; `transcribe` hands the batches out dynamically (a worker takes the next batch when it is done).
    Under rocprofv3 (`under_rocprof()`) the pipeline clamps itself to ONE worker: concurrent graph replays from several host threads
    aborted under the profiler in round 3 (cause not established, see `under_rocprof`); the kernels profiled are the same."""

    def __init__(self, encoder_buffer, decoder_buffer, config: dict, workers: int = 2, device: Optional[int] = None):
        import torch
        if workers < 1:
            raise ValueError("workers must be >= 1")
        if workers > 1 and under_rocprof():
            logger.warning(f"WhisperPipeline: rocprofv3 is loaded into this process -- running 1 worker instead of {workers} "
                           "(concurrent graph replays from several host threads abort under the profiler; the kernels are the same)")
            workers = 1
        self.device = torch.cuda.current_device() if device is None else device
        self.config = config
        enc0, dec0 = WhisperEncoderEngine(encoder_buffer, self.device), WhisperDecoderEngine(decoder_buffer, config, self.device)
        self.engines = [(enc0, dec0)] + [(enc0.clone(), dec0.clone()) for _ in range(workers - 1)]   # ONE copy of the weights on the device
        self.streams = [torch.cuda.Stream(device=self.device) for _ in range(workers)]

    def transcribe(self, mel_batches: Sequence[Any], gen_kwargs: Optional[Sequence[dict]] = None) -> List[Any]:
        """Encoder + greedy decode of every batch `[b_i, n_mels, 2*S]` (b_i <= 16 per engine call; larger ones are chunked by
        `generate`); returns the id tensors in the order of `mel_batches`.  `gen_kwargs[i]` are extra arguments of `generate` for
        batch i (max_length, force_eos_steps, ...).  Blocks until every batch is done."""
        import torch
        n = len(mel_batches)
        if gen_kwargs is not None and len(gen_kwargs) != n:
            raise ValueError("gen_kwargs needs one entry per batch")
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream(self.device))      # the caller's pending work on the inputs

        def setup(k: int):
            torch.cuda.set_device(self.device)
            self.streams[k].wait_event(ready)

        def item(k: int, i: int):
            enc, dec = self.engines[k]
            with torch.cuda.stream(self.streams[k]):
                return dec.generate(enc(mel_batches[i]), **(gen_kwargs[i] if gen_kwargs is not None else {}))

        return run_workers(n, len(self.engines), item, setup=setup, teardown=lambda k: self.streams[k].synchronize())


def _pipeline_transcribe_continuous(self, mels, slots: int = 8, chunk: int = 8, block: int = 64,
                                    force_eos_steps: Optional[Sequence[int]] = None, max_length=None) -> List[Any]:
    """Continuous decoding of `mels` ([N, n_mels, 2*S] CUDA tensor or list of rows) in ARRIVAL order: every worker takes blocks of `block`
    utterances and runs them through its own continuous stream (`transcribe_continuous`: `slots` rows busy, per-utterance stop, slots
    refilled on the device).  Returns N numpy id rows (start token through EOS) in input order."""
    import torch
    n = len(mels)
    blocks = [(a, min(a + block, n)) for a in range(0, n, block)]
    ready = torch.cuda.Event()
    ready.record(torch.cuda.current_stream(self.device))

    def setup(k: int):
        torch.cuda.set_device(self.device)
        self.streams[k].wait_event(ready)

    def item(k: int, j: int):
        a, b = blocks[j]
        enc, dec = self.engines[k]
        with torch.cuda.stream(self.streams[k]):
            return transcribe_continuous(enc, dec, mels[a:b], slots=slots, chunk=chunk, max_length=max_length,
                                         force_eos_steps=None if force_eos_steps is None else force_eos_steps[a:b])

    parts = run_workers(len(blocks), len(self.engines), item, setup=setup, teardown=lambda k: self.streams[k].synchronize())
    return [row for part in parts for row in part]


WhisperPipeline.transcribe_continuous = _pipeline_transcribe_continuous


def run_workers(n_items: int, n_workers: int, item, setup=None, teardown=None) -> List[Any]:
    """Host side of `WhisperPipeline.transcribe` (no GPU in here): `n_workers` threads take item indices 0 .. n_items-1 from a shared
    counter (a worker takes the next item when it is done with its own: dynamic, not round-robin), `item(worker, index)` produces
    result[index]; `setup(worker)` / `teardown(worker)` run once per thread.  The first exception stops the hand-out and is re-raised
    in the caller's thread after every worker has returned."""
    import itertools
    import threading
    results: List[Any] = [None] * n_items
    errors: List[BaseException] = []
    counter, lock = itertools.count(), threading.Lock()

    def work(k: int):
        try:
            if setup is not None:
                setup(k)
            while not errors:
                with lock:
                    i = next(counter)
                if i >= n_items:
                    break
                results[i] = item(k, i)
            if teardown is not None:
                teardown(k)
        except BaseException as exc:   # surfaced in the caller's thread below
            errors.append(exc)

    threads = [threading.Thread(target=work, args=(k,), daemon=True) for k in range(max(1, min(n_workers, n_items)))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errors:
        raise errors[0]
    return results
