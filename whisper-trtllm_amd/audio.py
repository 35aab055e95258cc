"""Log-mel front-end: constant tables (Hann window, DFT matrix, slaney mel filter bank) + the GPU front-end wrapper.

Replaces `hf_processor(audio, sampling_rate=..., return_tensors="pt").input_features` in the reference's timed loop
(examples/whisper/run.py:267), i.e. WhisperFeatureExtractor (feature_extraction_whisper.py:84-111, audio_utils.py:115-190,
206-264, 267-452).  The Hann window goes to the C-ABI in float64 (the DFT runs in fp64 on the matrix cores, like the reference's
float64 rfft), the mel filter bank in fp32; `dft_tables` is the fp32 real-DFT matrix kept for host-side checks."""
from __future__ import annotations

import ctypes

import numpy as np

from . import _lib

SAMPLING_RATE, N_FFT, HOP, N_MELS, CHUNK_SECONDS = 16000, 400, 160, 80, 30
N_SAMPLES = CHUNK_SECONDS * SAMPLING_RATE           # 480000
N_FRAMES = N_SAMPLES // HOP                         # 3000


def _hz_to_mel_slaney(f):
    f = np.asarray(f, dtype=np.float64)
    mel = 3.0 * f / 200.0
    logstep = 27.0 / np.log(6.4)
    return np.where(f >= 1000.0, 15.0 + np.log(np.maximum(f, 1e-300) / 1000.0) * logstep, mel)


def _mel_to_hz_slaney(m):
    m = np.asarray(m, dtype=np.float64)
    logstep = np.log(6.4) / 27.0
    return np.where(m >= 15.0, 1000.0 * np.exp(logstep * (m - 15.0)), 200.0 * m / 3.0)


def mel_filter_bank(n_freq: int = N_FFT // 2 + 1, n_mels: int = N_MELS, sr: int = SAMPLING_RATE, fmin: float = 0.0,
                    fmax: float = 8000.0) -> np.ndarray:
    """Slaney-scale, slaney-normalised triangular filters, float64 [n_freq, n_mels] (audio_utils.mel_filter_bank)."""
    fft_freqs = np.linspace(0, sr // 2, n_freq)
    edges = _mel_to_hz_slaney(np.linspace(_hz_to_mel_slaney(fmin), _hz_to_mel_slaney(fmax), n_mels + 2))
    diff = np.diff(edges)
    slopes = edges[None, :] - fft_freqs[:, None]
    down, up = -slopes[:, :-2] / diff[:-1], slopes[:, 2:] / diff[1:]
    fb = np.maximum(0.0, np.minimum(down, up))
    return fb * (2.0 / (edges[2:n_mels + 2] - edges[:n_mels]))[None, :]


def hann_window(n: int = N_FFT) -> np.ndarray:
    """Periodic Hann window (audio_utils.window_function(n, 'hann'): np.hanning(n + 1)[:-1])."""
    return np.hanning(n + 1)[:-1]


def dft_tables(n_fft: int = N_FFT):
    """Real-input DFT as a [ndft, n_fft] matrix: rows [0, n_bins) = cos, rows [ndft/2, ndft/2 + n_bins) = -sin."""
    n_bins = n_fft // 2 + 1
    half = (n_bins + 3) // 4 * 4 + (0 if (n_bins + 3) // 4 * 4 % 2 == 0 else 0)
    ndft = 2 * half
    k = np.arange(n_bins)[:, None] * np.arange(n_fft)[None, :]
    ang = 2.0 * np.pi * (k % n_fft) / n_fft
    m = np.zeros((ndft, n_fft), dtype=np.float64)
    m[:n_bins] = np.cos(ang)
    m[half:half + n_bins] = -np.sin(ang)
    return m.astype(np.float32), ndft


def valid_frames(mel) -> list:
    """Frames of real audio in each log-mel of a batch `[B, n_mels, T]` (torch tensor, any device): T minus the trailing run of
    frames identical to the last one.  A clip shorter than the 30 s window is zero-padded as a waveform
    (feature_extraction_whisper.py:229-250), so its padding frames are identical columns; a full window has no such run.
    Used as the length proxy of `sharding.length_sorted_batches`."""
    import torch
    if mel.dim() == 2:
        mel = mel[None]
    is_pad = (mel == mel[:, :, -1:]).all(dim=1)                  # [B, T]
    run = is_pad.flip(1).to(torch.int32).cumprod(dim=1).sum(dim=1)
    run = torch.where(run <= 1, torch.zeros_like(run), run)     # the last frame alone is not a padding run
    return [int(v) for v in (mel.shape[2] - run).cpu()]


class LogMelFrontend:
    """waveforms float32 [B, n] on the GPU (16 kHz mono) -> log-mel float32 [B, 80, 3000]; asynchronous."""

    def __init__(self, device=None):
        import torch
        self._lib = _lib.load()
        if not torch.cuda.is_available():
            raise RuntimeError("LogMelFrontend needs a ROCm GPU; there is no CPU execution path")
        dev = torch.cuda.current_device() if device is None else device
        n_bins = N_FFT // 2 + 1
        npw = (n_bins + 3) // 4 * 4
        filt = np.zeros((N_MELS, npw), dtype=np.float32)
        filt[:, :n_bins] = mel_filter_bank().T.astype(np.float32)
        win = np.ascontiguousarray(hann_window(), dtype=np.float64)   # the reference frames in float64 (audio_utils.py:399-401)
        self._handle = ctypes.c_void_p()
        _lib.check(self._lib.wt_logmel_create(dev, N_FFT, HOP, N_MELS, N_FRAMES, win.ctypes.data, filt.ctypes.data, npw,
                                              ctypes.byref(self._handle)), "wt_logmel_create")

    def __call__(self, waveforms):
        import torch
        if waveforms.dim() == 1:
            waveforms = waveforms[None]
        if waveforms.dtype != torch.float32 or not waveforms.is_cuda:
            raise ValueError("waveforms must be a float32 CUDA tensor [B, n_samples]")
        w = waveforms.contiguous()
        out = torch.empty(w.shape[0], N_MELS, N_FRAMES, dtype=torch.float32, device=w.device)
        stream = torch.cuda.current_stream().cuda_stream
        _lib.check(self._lib.wt_logmel_forward(self._handle, w.data_ptr(), w.shape[0], w.shape[1], out.data_ptr(),
                                               ctypes.c_void_p(stream)), "wt_logmel_forward")
        return out

    def __del__(self):
        try:
            if getattr(self, "_handle", None) is not None and self._handle.value:
                self._lib.wt_logmel_destroy(self._handle)
                self._handle = ctypes.c_void_p()
        except Exception:
            pass
