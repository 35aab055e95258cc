#!/usr/bin/env python3
"""Golden vectors for the log-mel front-end from the REFERENCE's bundled WhisperFeatureExtractor (build container only).

Run:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_frontend.py
Writes tests/golden/frontend.npz: the reference's mel filter bank (subsample + checksum) and its log-mel features for
seeded synthetic waveforms (`synthetic_waveform` below is re-implemented identically in tests/conftest.py)."""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF_SRC = "/root/reference/transformers/src"


def synthetic_waveform(seed: int, seconds: float) -> np.ndarray:
    """Noise + chirp + amplitude envelope at 16 kHz, float32 in [-1, 1]."""
    rng = np.random.default_rng(seed)
    n = int(16000 * seconds)
    t = np.arange(n) / 16000.0
    chirp = np.sin(2 * np.pi * (200.0 + 900.0 * t / max(seconds, 1e-3)) * t)
    env = 0.5 + 0.5 * np.sin(2 * np.pi * 0.7 * t + seed)
    return (0.6 * env * chirp + 0.05 * rng.standard_normal(n)).astype(np.float32)


CASES = [("full30s", 1, 30.0), ("short5s", 2, 5.3), ("long34s", 3, 34.0), ("silence_tail", 4, 12.0)]


def main():
    stub = types.ModuleType("transformers.dependency_versions_check")
    stub.dep_version_check = lambda *a, **k: None
    sys.modules["transformers.dependency_versions_check"] = stub
    sys.path.insert(0, REF_SRC)
    import transformers
    assert transformers.__version__ == "4.33.0.dev0" and transformers.__file__.startswith(REF_SRC)
    from transformers.models.whisper.feature_extraction_whisper import WhisperFeatureExtractor
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import cpu_ref

    fe = WhisperFeatureExtractor()
    out = {"mel_filters_sub": fe.mel_filters[::7, ::3].astype(np.float64), "mel_filters_sum": float(fe.mel_filters.sum()),
           "mel_filters_absmax": float(np.abs(fe.mel_filters).max())}
    assert np.abs(cpu_ref.whisper_mel_filters() - fe.mel_filters).max() < 1e-12
    for name, seed, seconds in CASES:
        wav = synthetic_waveform(seed, seconds)
        feats = fe(wav, sampling_rate=16000, return_tensors="np").input_features[0]          # [80, 3000]
        mine = cpu_ref.log_mel_spectrogram(wav)
        d = float(np.abs(mine - feats).max())
        print(f"[{name}] features {feats.shape} range [{feats.min():.3f}, {feats.max():.3f}] oracle vs reference {d:.2e}")
        assert feats.shape == (80, 3000) and d < 1e-5
        out[f"{name}_seed"], out[f"{name}_seconds"] = seed, seconds
        out[f"{name}_sub"] = feats[::5, ::37].astype(np.float32)
        out[f"{name}_mean"], out[f"{name}_max"], out[f"{name}_min"] = float(feats.mean()), float(feats.max()), float(feats.min())
        out[f"{name}_frames_head"] = feats[:, :4].astype(np.float32)
        out[f"{name}_frames_tail"] = feats[:, -4:].astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "frontend.npz"), **out)
    print("wrote frontend.npz", os.path.getsize(os.path.join(HERE, "frontend.npz")), "bytes")


if __name__ == "__main__":
    main()
