"""Seeded synthetic Whisper configs, weights and log-mel inputs.

There are no pretrained `whisper-*.en` checkpoints and no LibriSpeech data on
either box (SURVEY.md "Facts" item 3), so parity and throughput both run on
random-init weights of the real architecture.  Everything here is a pure
function of (config name, seed, tensor name): the oracle, the golden-vector
script (which loads these tensors into the reference's bundled HF model via
`load_state_dict`) and the HIP engine all see bit-identical fp32 weights.

Tensor names are the HF `state_dict` keys the reference's build scripts read
(build_encoder.py:71-91, build_decoder.py:71-101).
"""
from __future__ import annotations

import zlib
from typing import Dict, Iterator, Tuple

import numpy as np

# `.en` generation constants — SURVEY.md §8 note (d-1): recalled from the public
# hub configs, not present in the reference tree.
EN_SUPPRESS_TOKENS = [
    1, 2, 7, 8, 9, 10, 14, 25, 26, 27, 28, 29, 31, 58, 59, 60, 61, 62, 63, 90, 91, 92, 93, 357, 366, 438, 532, 685,
    705, 796, 930, 1058, 1220, 1267, 1279, 1303, 1343, 1377, 1391, 1635, 1782, 1875, 2162, 2361, 2488, 3467, 4008,
    4211, 4600, 4808, 5299, 5855, 6329, 7203, 9609, 9959, 10563, 10786, 11420, 11709, 11907, 13163, 13697, 13700,
    14808, 15306, 16410, 16791, 17992, 19203, 19510, 20724, 22305, 22935, 27007, 30109, 30420, 33409, 34949, 40283,
    40493, 40549, 47282, 49146, 50257, 50357, 50358, 50359, 50360, 50361,
]


def _en(d_model, layers, heads, ffn):
    return dict(
        d_model=d_model, num_mel_bins=80, max_source_positions=1500, max_target_positions=448,
        encoder_layers=layers, decoder_layers=layers, encoder_attention_heads=heads, decoder_attention_heads=heads,
        encoder_ffn_dim=ffn, decoder_ffn_dim=ffn, vocab_size=51864, activation_function="gelu",
        scale_embedding=False, pad_token_id=50256, eos_token_id=50256, bos_token_id=50256,
        decoder_start_token_id=50257, forced_decoder_ids=[[1, 50362]], forced_bos_token_id=None,
        begin_suppress_tokens=[220, 50256], suppress_tokens=list(EN_SUPPRESS_TOKENS), max_length=448,
    )


def _toy(d_model=128, layers=2, heads=2, ffn=256, vocab=512, src=1500, tgt=64, max_length=24):
    return dict(
        d_model=d_model, num_mel_bins=80, max_source_positions=src, max_target_positions=tgt,
        encoder_layers=layers, decoder_layers=layers, encoder_attention_heads=heads, decoder_attention_heads=heads,
        encoder_ffn_dim=ffn, decoder_ffn_dim=ffn, vocab_size=vocab, activation_function="gelu",
        scale_embedding=False, pad_token_id=vocab - 3, eos_token_id=vocab - 3, bos_token_id=vocab - 3,
        decoder_start_token_id=vocab - 4, forced_decoder_ids=[[1, vocab - 2]], forced_bos_token_id=None,
        begin_suppress_tokens=[220, vocab - 1], suppress_tokens=[1, 2, 7], max_length=max_length,
    )


CONFIGS = {
    "whisper-tiny.en": _en(384, 4, 6, 1536),
    "whisper-base.en": _en(512, 6, 8, 2048),
    "whisper-small.en": _en(768, 12, 12, 3072),
    "whisper-medium.en": _en(1024, 24, 16, 4096),
    # toy shapes for second-scale CPU tests; head_dim stays 64 like every Whisper size
    "toy": _toy(),                                   # SURVEY §8(c) toy: full 1500-frame memory
    "toy-short": _toy(src=96, tgt=40, max_length=20),  # 192 mel frames, 96-frame memory
    "toy-wide": _toy(d_model=192, heads=3, ffn=320, vocab=1000, src=160, tgt=48, max_length=32, layers=3),
}


def get_config(name: str) -> dict:
    if name not in CONFIGS:
        raise KeyError(f"unknown synthetic whisper config {name!r}; have {sorted(CONFIGS)}")
    cfg = dict(CONFIGS[name])
    cfg["name"] = name
    return cfg


def weight_specs(cfg: dict) -> Iterator[Tuple[str, Tuple[int, ...], str]]:
    """Yield (hf_key, shape, kind) for every tensor the build scripts read."""
    d, nm = cfg["d_model"], cfg["num_mel_bins"]
    yield "model.encoder.conv1.weight", (d, nm, 3), "conv"
    yield "model.encoder.conv1.bias", (d,), "bias"
    yield "model.encoder.conv2.weight", (d, d, 3), "conv"
    yield "model.encoder.conv2.bias", (d,), "bias"
    yield "model.encoder.embed_positions.weight", (cfg["max_source_positions"], d), "pos"
    for i in range(cfg["encoder_layers"]):
        p = f"model.encoder.layers.{i}."
        f = cfg["encoder_ffn_dim"]
        for nme in ("q_proj", "k_proj", "v_proj", "out_proj"):
            yield p + f"self_attn.{nme}.weight", (d, d), "linear"
            if nme != "k_proj":
                yield p + f"self_attn.{nme}.bias", (d,), "bias"
        yield p + "self_attn_layer_norm.weight", (d,), "ln_w"
        yield p + "self_attn_layer_norm.bias", (d,), "ln_b"
        yield p + "fc1.weight", (f, d), "linear"
        yield p + "fc1.bias", (f,), "bias"
        yield p + "fc2.weight", (d, f), "linear"
        yield p + "fc2.bias", (d,), "bias"
        yield p + "final_layer_norm.weight", (d,), "ln_w"
        yield p + "final_layer_norm.bias", (d,), "ln_b"
    yield "model.encoder.layer_norm.weight", (d,), "ln_w"
    yield "model.encoder.layer_norm.bias", (d,), "ln_b"
    yield "model.decoder.embed_tokens.weight", (cfg["vocab_size"], d), "embed"
    yield "model.decoder.embed_positions.weight", (cfg["max_target_positions"], d), "pos"
    for i in range(cfg["decoder_layers"]):
        p = f"model.decoder.layers.{i}."
        f = cfg["decoder_ffn_dim"]
        for attn in ("self_attn", "encoder_attn"):
            for nme in ("q_proj", "k_proj", "v_proj", "out_proj"):
                yield p + f"{attn}.{nme}.weight", (d, d), "linear"
                if nme != "k_proj":
                    yield p + f"{attn}.{nme}.bias", (d,), "bias"
            yield p + f"{attn}_layer_norm.weight", (d,), "ln_w"
            yield p + f"{attn}_layer_norm.bias", (d,), "ln_b"
        yield p + "fc1.weight", (f, d), "linear"
        yield p + "fc1.bias", (f,), "bias"
        yield p + "fc2.weight", (d, f), "linear"
        yield p + "fc2.bias", (d,), "bias"
        yield p + "final_layer_norm.weight", (d,), "ln_w"
        yield p + "final_layer_norm.bias", (d,), "ln_b"
    yield "model.decoder.layer_norm.weight", (d,), "ln_w"
    yield "model.decoder.layer_norm.bias", (d,), "ln_b"
    # proj_out.weight is tied to model.decoder.embed_tokens.weight (modeling_whisper.py:1335)


def _tensor(name: str, shape, kind: str, seed: int) -> np.ndarray:
    rng = np.random.default_rng([seed, zlib.crc32(name.encode())])
    if kind == "linear":
        std = 1.6 / np.sqrt(shape[1])
    elif kind == "conv":
        std = 1.6 / np.sqrt(shape[1] * shape[2])
    elif kind == "embed":
        std = 0.12
    elif kind == "pos":
        std = 0.25
    elif kind == "bias":
        std = 0.1
    elif kind == "ln_w":
        return (1.0 + 0.1 * rng.standard_normal(shape, dtype=np.float32)).astype(np.float32)
    elif kind == "ln_b":
        std = 0.1
    else:
        raise ValueError(kind)
    return (std * rng.standard_normal(shape, dtype=np.float32)).astype(np.float32)


def make_weights(cfg: dict, seed: int = 0) -> Dict[str, np.ndarray]:
    """fp32 weights keyed by HF state_dict name; `proj_out.weight` aliases the token embedding."""
    out = {}
    for name, shape, kind in weight_specs(cfg):
        out[name] = _tensor(name, shape, kind, seed)
    out["proj_out.weight"] = out["model.decoder.embed_tokens.weight"]
    return out


def make_mel(cfg: dict, index: int, batch: int = 1) -> np.ndarray:
    """Synthetic log-mel `[batch, n_mels, 2*max_source_positions]`, U(-1,1)."""
    frames = 2 * cfg["max_source_positions"]
    return np.stack([
        np.random.default_rng(1000 + index + b).uniform(-1.0, 1.0, (cfg["num_mel_bins"], frames)).astype(np.float32)
        for b in range(batch)
    ])


# ---- variable-length workload (bench.py `value_varlen`): LibriSpeech-like utterance durations and transcript lengths ------------------
PAD_MEL = -1.5   # value of the padding frames behind the audio (below every "speech" value; real features: (max - 8 + 4) / 4, constant)


def librispeech_like_lengths(n: int, seed: int = 0, max_length: int = 448):
    """(durations_s, eos_steps) for `n` utterances.  MODELLED on the published shape of LibriSpeech test-clean (2620 utterances,
    1.3-35 s, mean 7.4 s; no copy of the dataset exists on either box): duration ~ lognormal(ln 6.2, 0.62) clipped to [1.3, 30] s,
    transcript = 3.4 BPE tokens per second + the two prompt positions; `eos_steps[i]` is the 0-based decoder step whose token is EOS."""
    rng = np.random.default_rng([seed, 0x11b5])
    dur = np.clip(rng.lognormal(np.log(6.2), 0.62, n), 1.3, 30.0)
    steps = np.clip(np.rint(3.4 * dur).astype(np.int64) + 2, 2, max_length - 2)
    return dur.astype(np.float64), [int(s) for s in steps]


def make_mel_padded(cfg: dict, index: int, duration_s: float) -> np.ndarray:
    """One synthetic log-mel `[n_mels, frames]`: U(-1,1) over the first duration_s * 100 frames, PAD_MEL behind them -- the shape
    the front-end gives a clip shorter than the 30 s window (feature_extraction_whisper.py:229-250 pads the WAVEFORM with zeros)."""
    frames = 2 * cfg["max_source_positions"]
    valid = max(1, min(frames, int(round(duration_s * 100.0))))
    mel = np.full((cfg["num_mel_bins"], frames), PAD_MEL, dtype=np.float32)
    mel[:, :valid] = np.random.default_rng(5000 + index).uniform(-1.0, 1.0, (cfg["num_mel_bins"], valid)).astype(np.float32)
    return mel
