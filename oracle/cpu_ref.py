"""CPU oracle for the Whisper encoder-decoder greedy path.  TEST INFRASTRUCTURE ONLY.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import this module; the product (`whisper-trtllm_amd/`) never does and fails
loudly when its HIP library is missing.

This is a torch-CPU fp32, op-for-op restatement of the reference's bundled
HuggingFace path (the oracle `BASELINE.json.north_star` names) plus the
TensorRT-LLM engine-surface semantics layered on top of it.  Citations use
  HF/ = /root/reference/transformers/src/transformers/
  TL/ = /root/reference/tensorrt_llm_july-release-v1/

Parity pinning: `tests/golden/*.npz` were produced by `tests/golden/make_golden.py`,
which imports the bundled HF model in the build container, loads the same seeded
weights and records its outputs; `tests/test_oracle.py` checks this file against
those vectors (logits <= 1e-4, token ids exact).  The TensorRT engines themselves
cannot be built here (closed TensorRT 9) — for that surface the oracle follows
`TL/tensorrt_llm/models/whisper/model.py` as text: parity unpinned for the
engine-only quirks (mask-shape cache gating at intermediate lengths).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Weights = Dict[str, torch.Tensor]


def to_torch(weights: Dict[str, np.ndarray]) -> Weights:
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in weights.items()}


# ----------------------------------------------------------------------------- fp16 engines (--engine_precision float16)
# The reference builds fp16 engines with TensorRT (TL/examples/whisper/build_{en,de}coder.py:25,62, TL/tensorrt_llm/builder.py:55)
# and holds NO fp16 fixture: parity of the fp16 numbers themselves is unpinned.  What CAN be pinned is the arithmetic: an fp16
# engine of this repo is the fp32 model evaluated on fp16-ROUNDED weights, with a few tensors rounded to fp16 where the engine
# stores them (the encoder's GEMM-input activations; the decoder's resident K/V caches and the encoder memory fed to the cross-K/V
# GEMM), all sums / LayerNorm / softmax / residual in fp32 (TL model.py:292-295 forces fp32 scores as well).  The functions below
# restate exactly that in fp32 torch, so tests compare at ~1e-3 of the logits instead of a percent-of-range tolerance.
def _r16(t: torch.Tensor) -> torch.Tensor:
    return t.half().float()


def fp16_engine_weights(weights: Dict[str, np.ndarray], encoder: bool = False, decoder: bool = False) -> Dict[str, np.ndarray]:
    """The weights an fp16 engine computes with: every tensor `builder.py` stores as IEEE half (conv / linear weight matrices, the
    token table tied to the vocabulary projection) rounded to fp16 and widened back; biases, LayerNorm parameters and both position
    tables unchanged."""
    out = {}
    for k, v in weights.items():
        side = "encoder" if k.startswith("model.encoder.") else "decoder"   # proj_out.weight is the decoder's
        is_matrix = k.endswith(".weight") and v.ndim >= 2 and "embed_positions" not in k
        if is_matrix and ((side == "encoder" and encoder) or (side == "decoder" and decoder)):
            out[k] = v.astype(np.float16).astype(np.float32)
        else:
            out[k] = v
    if "proj_out.weight" in weights and weights["proj_out.weight"] is weights.get("model.decoder.embed_tokens.weight"):
        out["proj_out.weight"] = out["model.decoder.embed_tokens.weight"]   # keep the tie
    return out


# ----------------------------------------------------------------------------- encoder
def _encoder_attention(x: torch.Tensor, W: Weights, p: str, n_heads: int, fp16_engine: bool = False) -> torch.Tensor:
    """HF/models/whisper/modeling_whisper.py:569-593 (WhisperEncoderAttention.forward).

    fp16_engine (this repo's fp16 encoder engine; x arrives fp16-rounded): q, k, v are stored as fp16, the scores and the softmax
    are fp32 (TL model.py:292-295), the probabilities enter the P.V product as fp16 while their row sum stays fp32, and the context
    is stored as fp16.  (The engine rounds exp(s - running max) tile by tile and rescales in fp32; rounding exp(s - max) once, as
    here, has the same error size but not the same bits -- tests compare at a few 1e-4 of the range, not bit for bit.)"""
    r = _r16 if fp16_engine else (lambda t: t)
    B, T, D = x.shape
    dh = D // n_heads
    q = r(F.linear(x, W[p + "q_proj.weight"], W[p + "q_proj.bias"])) * (dh ** -0.5)  # :572 q scaled BEFORE QK^T (exact in fp16)
    k = r(F.linear(x, W[p + "k_proj.weight"]))                                    # k_proj has no bias (:547)
    v = r(F.linear(x, W[p + "v_proj.weight"], W[p + "v_proj.bias"]))
    sh = lambda t: t.view(B, T, n_heads, dh).transpose(1, 2)
    q, k, v = sh(q), sh(k), sh(v)
    if fp16_engine:
        sc = q @ k.transpose(-1, -2)
        e = torch.exp(sc - sc.max(dim=-1, keepdim=True).values)
        ctx = r((_r16(e) @ v) / e.sum(dim=-1, keepdim=True))
    else:
        att = torch.softmax(q @ k.transpose(-1, -2), dim=-1)                       # :581-583, no mask is ever added
        ctx = att @ v
    ctx = ctx.transpose(1, 2).reshape(B, T, D)
    return F.linear(ctx, W[p + "out_proj.weight"], W[p + "out_proj.bias"])


def encoder_conv_frontend(W: Weights, cfg: dict, mel: torch.Tensor, fp16_engine: bool = False) -> torch.Tensor:
    """HF modeling_whisper.py:992-997: conv1+GELU(erf), conv2(stride 2)+GELU, permute, + embed_positions.
    fp16_engine: the mel and conv1's output are the fp16 A operands of the two implicit GEMMs."""
    r = _r16 if fp16_engine else (lambda t: t)
    x = r(F.gelu(F.conv1d(r(mel), W["model.encoder.conv1.weight"], W["model.encoder.conv1.bias"], padding=1)))
    x = F.gelu(F.conv1d(x, W["model.encoder.conv2.weight"], W["model.encoder.conv2.bias"], stride=2, padding=1))
    return x.permute(0, 2, 1) + W["model.encoder.embed_positions.weight"]


def encoder_layer(W: Weights, cfg: dict, i: int, h: torch.Tensor, fp16_engine: bool = False) -> torch.Tensor:
    """HF modeling_whisper.py:632-641 (pre-LN attention + pre-LN FFN).  fp16_engine: LayerNorm outputs and the GELU output are the
    fp16 A operands of the next GEMM; the residual stream stays fp32."""
    r = _r16 if fp16_engine else (lambda t: t)
    p = f"model.encoder.layers.{i}."
    D = cfg["d_model"]
    res = h
    x = r(F.layer_norm(h, (D,), W[p + "self_attn_layer_norm.weight"], W[p + "self_attn_layer_norm.bias"], 1e-5))
    h = res + _encoder_attention(x, W, p + "self_attn.", cfg["encoder_attention_heads"], fp16_engine)
    res = h
    x = r(F.layer_norm(h, (D,), W[p + "final_layer_norm.weight"], W[p + "final_layer_norm.bias"], 1e-5))
    x = r(F.gelu(F.linear(x, W[p + "fc1.weight"], W[p + "fc1.bias"])))
    return res + F.linear(x, W[p + "fc2.weight"], W[p + "fc2.bias"])


def encoder_forward(W: Weights, cfg: dict, mel: torch.Tensor, fp16_engine: bool = False) -> torch.Tensor:
    """HF WhisperEncoder.forward modeling_whisper.py:992-1011 == TL model.py:90-111 (with erf GELU).

    mel f32 [B, 80, 2*max_source_positions] -> f32 [B, max_source_positions, d_model].
    fp16_engine: the arithmetic of this repo's fp16 encoder engine (pass `fp16_engine_weights(.., encoder=True)` as W): fp16 GEMM
    operands, fp32 accumulation / residual stream / LayerNorm / softmax."""
    h = encoder_conv_frontend(W, cfg, mel, fp16_engine)
    for i in range(cfg["encoder_layers"]):
        h = encoder_layer(W, cfg, i, h, fp16_engine)
    D = cfg["d_model"]
    return F.layer_norm(h, (D,), W["model.encoder.layer_norm.weight"], W["model.encoder.layer_norm.bias"], 1e-5)


# ----------------------------------------------------------------------------- decoder (HF semantics)
def _decoder_attention(x, W, p, n_heads, kv_states=None, past=None, kv_half=False):
    """HF modeling_whisper.py:468-526 (WhisperDecoderAttention.forward), four branches.
    kv_half (fp16 engines, fast path): new K/V rows are rounded to fp16 where the engine's resident caches store them."""
    B, T, D = x.shape
    dh = D // n_heads
    sh = lambda t: t.view(B, -1, n_heads, dh).transpose(1, 2)
    q = F.linear(x, W[p + "q_proj.weight"], W[p + "q_proj.bias"]) * (dh ** -0.5)  # :472
    if kv_states is not None and past is not None and past[0].shape[2] == kv_states.shape[1]:
        k, v = past                                                               # :474-481 reuse cross K/V
    elif kv_states is not None:                                                   # :484-486 first-step cross K/V
        k = sh(F.linear(kv_states, W[p + "k_proj.weight"]))
        v = sh(F.linear(kv_states, W[p + "v_proj.weight"], W[p + "v_proj.bias"]))
        if kv_half:
            k, v = _r16(k), _r16(v)
    else:                                                                         # :490-503 self attention
        k = sh(F.linear(x, W[p + "k_proj.weight"]))
        v = sh(F.linear(x, W[p + "v_proj.weight"], W[p + "v_proj.bias"]))
        if kv_half:
            k, v = _r16(k), _r16(v)
        if past is not None:
            k = torch.cat([past[0], k], dim=2)
            v = torch.cat([past[1], v], dim=2)
    att = torch.softmax(sh(q) @ k.transpose(-1, -2), dim=-1)                      # :513-515, no mask
    ctx = (att @ v).transpose(1, 2).reshape(B, T, D)
    return F.linear(ctx, W[p + "out_proj.weight"], W[p + "out_proj.bias"]), (k, v)


def decoder_forward(W: Weights, cfg: dict, input_ids: torch.Tensor, enc_out: torch.Tensor, past=None, fp16_engine: bool = False):
    """HF WhisperDecoder.forward :1143-1185 + proj_out :1433.
    fp16_engine: the arithmetic of this repo's fp16 decoder engine on its fast path (pass `fp16_engine_weights(.., decoder=True)` as W):
    the encoder memory is rounded to fp16 for the cross-K/V projection, K/V rows are rounded to fp16 where the resident caches store them.

    input_ids i64 [B, T] (T=1 on every greedy step), enc_out [B, S, D],
    past: tuple over layers of (self_k, self_v, cross_k, cross_v) each [B,H,*,64] or None.
    Returns (logits [B, T, V], present)."""
    D = cfg["d_model"]
    H = cfg["decoder_attention_heads"]
    past_len = past[0][0].shape[2] if past is not None else 0                      # :1147
    h = F.embedding(input_ids, W["model.decoder.embed_tokens.weight"])             # :1149 (embed_scale never applied)
    h = h + W["model.decoder.embed_positions.weight"][past_len:past_len + input_ids.shape[1]]  # :1154, :308
    present = []
    enc_kv = _r16(enc_out) if fp16_engine else enc_out
    for i in range(cfg["decoder_layers"]):
        p = f"model.decoder.layers.{i}."
        lp = past[i] if past is not None else None
        r = h                                                                      # :710-751 WhisperDecoderLayer
        x = F.layer_norm(h, (D,), W[p + "self_attn_layer_norm.weight"], W[p + "self_attn_layer_norm.bias"], 1e-5)
        a, (sk, sv) = _decoder_attention(x, W, p + "self_attn.", H, None, lp[:2] if lp is not None else None, kv_half=fp16_engine)
        h = r + a
        r = h
        x = F.layer_norm(h, (D,), W[p + "encoder_attn_layer_norm.weight"], W[p + "encoder_attn_layer_norm.bias"], 1e-5)
        a, (ck, cv) = _decoder_attention(x, W, p + "encoder_attn.", H, enc_kv, lp[2:] if lp is not None else None, kv_half=fp16_engine)
        h = r + a
        r = h
        x = F.layer_norm(h, (D,), W[p + "final_layer_norm.weight"], W[p + "final_layer_norm.bias"], 1e-5)
        x = F.gelu(F.linear(x, W[p + "fc1.weight"], W[p + "fc1.bias"]))
        h = r + F.linear(x, W[p + "fc2.weight"], W[p + "fc2.bias"])
        present.append((sk, sv, ck, cv))
    h = F.layer_norm(h, (D,), W["model.decoder.layer_norm.weight"], W["model.decoder.layer_norm.bias"], 1e-5)
    logits = F.linear(h, W["proj_out.weight"])                                      # :1433, tied, no bias
    return logits, tuple(present)


# ----------------------------------------------------------------------------- decoder (engine surface)
def engine_decoder_step(W: Weights, cfg: dict, data: torch.Tensor, enc_out: torch.Tensor,
                        self_past_key: torch.Tensor, self_past_value: torch.Tensor,
                        cross_past_key: torch.Tensor, cross_past_value: torch.Tensor,
                        m_s: int, m_c: int, fp16_engine: bool = False):
    """The TensorRT-LLM WhisperDecoder engine contract (TL model.py:407-470), batch 1.

    data i32 [1,1]; caches [L,H,s,64] / [L,H,S_enc,64]; m_s/m_c = LENGTHS of the two mask inputs (values unused).
      position row         = m_s - 1                               (model.py:424)
      self  cache_len      = min(m_s - 1, s)                       (model.py:278)
      cross cache_len  c   = m_c - 1 ; cur = proj(enc[0 : S-c])    (model.py:264-269, slice starts at 0)
    Follows HF numerics (erf GELU, q pre-scaled) where the two differ.
    fp16_engine: this repo's fp16 decoder engine behind the Session surface (W = fp16_engine_weights(.., decoder=True)): the caches
    are the caller's f32 tensors (TL model.py:464-468 casts them to f32 in fp16 builds too), so only the encoder rows fed to the
    cross-K/V GEMM are rounded to fp16.
    Returns (logits [1,1,V], next_self_keys, next_self_values, next_cross_keys, next_cross_values)."""
    D, H, L = cfg["d_model"], cfg["decoder_attention_heads"], cfg["decoder_layers"]
    S = cfg["max_source_positions"]
    dh = D // H
    sh = lambda t: t.view(1, -1, H, dh).transpose(1, 2)
    h = F.embedding(data.long(), W["model.decoder.embed_tokens.weight"])
    h = h + W["model.decoder.embed_positions.weight"][m_s - 1:m_s - 1 + data.shape[1]]
    nsk, nsv, nck, ncv = [], [], [], []

    def attend(x, p, k, v):
        q = sh(F.linear(x, W[p + "q_proj.weight"], W[p + "q_proj.bias"]) * (dh ** -0.5))
        att = torch.softmax(q @ k.transpose(-1, -2), dim=-1)
        ctx = (att @ v).transpose(1, 2).reshape(1, -1, D)
        return F.linear(ctx, W[p + "out_proj.weight"], W[p + "out_proj.bias"])

    for i in range(L):
        p = f"model.decoder.layers.{i}."
        r = h
        x = F.layer_norm(h, (D,), W[p + "self_attn_layer_norm.weight"], W[p + "self_attn_layer_norm.bias"], 1e-5)
        c = min(m_s - 1, self_past_key.shape[2])
        k = torch.cat([self_past_key[i:i + 1, :, :c], sh(F.linear(x, W[p + "self_attn.k_proj.weight"]))], dim=2)
        v = torch.cat([self_past_value[i:i + 1, :, :c],
                       sh(F.linear(x, W[p + "self_attn.v_proj.weight"], W[p + "self_attn.v_proj.bias"]))], dim=2)
        h = r + attend(x, p + "self_attn.", k, v)
        nsk.append(k); nsv.append(v)
        r = h
        x = F.layer_norm(h, (D,), W[p + "encoder_attn_layer_norm.weight"], W[p + "encoder_attn_layer_norm.bias"], 1e-5)
        c = m_c - 1
        cur = enc_out[:, 0:S - c]
        if fp16_engine:
            cur = _r16(cur)
        k = torch.cat([cross_past_key[i:i + 1, :, :c], sh(F.linear(cur, W[p + "encoder_attn.k_proj.weight"]))], dim=2)
        v = torch.cat([cross_past_value[i:i + 1, :, :c],
                       sh(F.linear(cur, W[p + "encoder_attn.v_proj.weight"], W[p + "encoder_attn.v_proj.bias"]))], dim=2)
        h = r + attend(x, p + "encoder_attn.", k, v)
        nck.append(k); ncv.append(v)
        r = h
        x = F.layer_norm(h, (D,), W[p + "final_layer_norm.weight"], W[p + "final_layer_norm.bias"], 1e-5)
        x = F.gelu(F.linear(x, W[p + "fc1.weight"], W[p + "fc1.bias"]))
        h = r + F.linear(x, W[p + "fc2.weight"], W[p + "fc2.bias"])
    h = F.layer_norm(h, (D,), W["model.decoder.layer_norm.weight"], W["model.decoder.layer_norm.bias"], 1e-5)
    logits = F.linear(h, W["proj_out.weight"])
    return logits, torch.cat(nsk, 0), torch.cat(nsv, 0), torch.cat(nck, 0), torch.cat(ncv, 0)


# ----------------------------------------------------------------------------- generation
def apply_logits_processors(cfg: dict, cur_len: int, prompt_len: int, scores: torch.Tensor) -> torch.Tensor:
    """run.py:150-162 == HF generation/utils.py:890-902; classes HF generation/logits_process.py:1281-1328.

    Order Suppress -> SuppressAtBegin -> Force; `cur_len` = input_ids.shape[1] at the call."""
    scores = scores.clone()
    scores[:, list(cfg["suppress_tokens"])] = -float("inf")                        # :1308-1310
    begin_index = prompt_len if cfg.get("forced_bos_token_id") is None else prompt_len + 1
    begin_index += cfg["forced_decoder_ids"][-1][0]                                # utils.py:897-899
    if cur_len == begin_index:                                                     # :1293-1295
        scores[:, list(cfg["begin_suppress_tokens"])] = -float("inf")
    forced = dict(cfg["forced_decoder_ids"]).get(cur_len, None)                    # :1321-1327
    if forced is not None:
        scores[:, :] = -float("inf")
        scores[:, forced] = 0
    return scores


def greedy_search(W: Weights, cfg: dict, enc_out: torch.Tensor, max_length: Optional[int] = None,
                  return_logits: bool = False, force_eos_at: Optional[int] = None, fp16_engine: bool = False):
    """run.py:171-227 == HF generation/utils.py:1474-1529 with the HF decoder as the model.

    Starts from [[decoder_start_token_id]] per row (run.py:273); stops when every row has emitted EOS or
    len >= max_length (MaxLengthCriteria, HF stopping_criteria.py:61-70).  `force_eos_at=n` (bench only)
    makes step n emit EOS for every row, emulating a LibriSpeech-length transcript on random weights; a sequence of B
    entries does so per row (entry < 0: never) -- the variable-length workload of bench.py."""
    B = enc_out.shape[0]
    max_length = cfg["max_length"] if max_length is None else max_length
    eos, pad = cfg["eos_token_id"], cfg["pad_token_id"]
    ids = torch.full((B, 1), cfg["decoder_start_token_id"], dtype=torch.long)
    unfinished = torch.ones(B, dtype=torch.long)
    past = None
    all_logits: List[torch.Tensor] = []
    step = 0
    while True:
        logits, past = decoder_forward(W, cfg, ids[:, -1:], enc_out, past, fp16_engine=fp16_engine)
        nxt_logits = logits[:, -1, :]
        if return_logits:
            all_logits.append(nxt_logits.clone())
        scores = apply_logits_processors(cfg, ids.shape[1], 1, nxt_logits)
        nxt = torch.argmax(scores, dim=-1)
        if force_eos_at is not None:
            if isinstance(force_eos_at, int):
                if step == force_eos_at:
                    nxt = torch.full_like(nxt, eos)
            else:
                hit = torch.tensor([int(f) == step for f in force_eos_at], dtype=torch.bool)
                nxt = torch.where(hit, torch.full_like(nxt, eos), nxt)
        nxt = nxt * unfinished + pad * (1 - unfinished)                            # utils.py:1509
        ids = torch.cat([ids, nxt[:, None]], dim=-1)
        unfinished = unfinished * (nxt != eos).long()                              # utils.py:1514
        step += 1
        if unfinished.max() == 0 or ids.shape[1] >= max_length:
            break
    if return_logits:
        return ids, torch.stack(all_logits, dim=1)
    return ids


def transcribe(W: Weights, cfg: dict, mel: torch.Tensor, **kw):
    """Encoder + greedy decode == `hf_model.generate(mel)` in run.py:305-306."""
    return greedy_search(W, cfg, encoder_forward(W, cfg, mel), **kw)


# ----------------------------------------------------------------------------- log-mel front-end
def _hertz_to_mel_slaney(freq):
    """HF audio_utils.py:25-57 (mel_scale='slaney')."""
    freq = np.asarray(freq, dtype=np.float64)
    mels = 3.0 * freq / 200.0
    logstep = 27.0 / np.log(6.4)
    with np.errstate(divide="ignore"):
        return np.where(freq >= 1000.0, 15.0 + np.log(np.maximum(freq, 1e-300) / 1000.0) * logstep, mels)


def _mel_to_hertz_slaney(mels):
    """HF audio_utils.py:59-90."""
    mels = np.asarray(mels, dtype=np.float64)
    logstep = np.log(6.4) / 27.0
    return np.where(mels >= 15.0, 1000.0 * np.exp(logstep * (mels - 15.0)), 200.0 * mels / 3.0)


def whisper_mel_filters(n_freq: int = 201, n_mels: int = 80, sr: int = 16000) -> np.ndarray:
    """HF audio_utils.mel_filter_bank(201, 80, 0, 8000, 16000, norm='slaney', mel_scale='slaney') :115-190 -> [n_freq, n_mels]."""
    fft_freqs = np.linspace(0, sr // 2, n_freq)
    mel_freqs = np.linspace(_hertz_to_mel_slaney(0.0), _hertz_to_mel_slaney(8000.0), n_mels + 2)
    filter_freqs = _mel_to_hertz_slaney(mel_freqs)
    filter_diff = np.diff(filter_freqs)
    slopes = np.expand_dims(filter_freqs, 0) - np.expand_dims(fft_freqs, 1)
    down = -slopes[:, :-2] / filter_diff[:-1]
    up = slopes[:, 2:] / filter_diff[1:]
    fb = np.maximum(np.zeros(1), np.minimum(down, up))
    return fb * np.expand_dims(2.0 / (filter_freqs[2:n_mels + 2] - filter_freqs[:n_mels]), 0)


def log_mel_spectrogram(waveform: np.ndarray) -> np.ndarray:
    """HF WhisperFeatureExtractor: pad/trim to 30 s (feature_extraction_whisper.py:229-237), then
    _np_extract_fbank_features :94-111 over audio_utils.spectrogram :267-452 (float64 frames, rfft stored as
    complex64, |.|^2, mel, max(1e-10), log10, float32), drop last frame, clamp to max-8, (x+4)/4.
    waveform float32 [n] at 16 kHz -> float32 [80, 3000]."""
    n_fft, hop, n_samples = 400, 160, 480000
    x = np.zeros(n_samples, dtype=np.float32)
    w = np.asarray(waveform, dtype=np.float32)[:n_samples]
    x[:len(w)] = w
    x = np.pad(x, [(n_fft // 2, n_fft // 2)], mode="reflect").astype(np.float64)
    window = np.hanning(n_fft + 1)[:-1].astype(np.float64)
    num_frames = int(1 + np.floor((x.size - n_fft) / hop))
    idx = np.arange(n_fft)[None, :] + hop * np.arange(num_frames)[:, None]
    spec = np.fft.rfft(x[idx] * window[None, :], axis=-1).astype(np.complex64)
    power = np.abs(spec, dtype=np.float64) ** 2.0
    mel = np.maximum(1e-10, np.dot(whisper_mel_filters().T, power.T))
    log_spec = np.asarray(np.log10(mel), np.float32)[:, :-1]
    log_spec = np.maximum(log_spec, log_spec.max() - 8.0)
    return (log_spec + 4.0) / 4.0
