#!/usr/bin/env python3
"""Build `librispeech.cache` — a pickled list of (log-mel float32 [80, 3000], transcript) pairs — from a LOCAL LibriSpeech-style
directory, the artefact the reference's examples/whisper/get_LibriSpeech.py produces for cal_wer.py (:31-39).

Differences from the reference script, both forced by the boxes this runs on: nothing is downloaded (there is no network; the
reference calls torchaudio.datasets.LIBRISPEECH(download=True)), and the log-mel features come from this package's GPU front-end
(`whisper_trtllm_amd.audio.LogMelFrontend`, the WhisperFeatureExtractor algorithm of run.py:267) instead of
`whisper.log_mel_spectrogram` on the CPU.  Audio: 16 kHz mono; `.wav` (16-bit PCM) is read with the standard library, `.flac`
needs `soundfile` or `torchaudio` to be importable.  Layout: every `*.trans.txt` lists `<utterance-id> <TRANSCRIPT>` lines and
the audio of an utterance is `<utterance-id>.flac|.wav` next to it (LibriSpeech's own layout)."""
import argparse
import glob
import os
import pickle
import wave

import numpy as np
import torch

from _common import ROOT  # noqa: F401

from whisper_trtllm_amd.audio import LogMelFrontend

SAMPLE_RATE, N_SAMPLES = 16000, 480000


def read_audio(path: str) -> np.ndarray:
    if path.endswith(".wav"):
        with wave.open(path, "rb") as f:
            if f.getframerate() != SAMPLE_RATE or f.getnchannels() != 1 or f.getsampwidth() != 2:
                raise SystemExit(f"{path}: need 16 kHz mono 16-bit PCM")
            return np.frombuffer(f.readframes(f.getnframes()), dtype="<i2").astype(np.float32) / 32768.0
    try:
        import soundfile
        audio, rate = soundfile.read(path, dtype="float32")
    except ImportError:
        try:
            import torchaudio
            t, rate = torchaudio.load(path)
            audio = t.flatten().numpy()
        except ImportError:
            raise SystemExit(f"{path}: reading FLAC needs `soundfile` or `torchaudio`; convert to 16 kHz mono .wav instead")
    assert rate == SAMPLE_RATE, f"{path}: sample rate {rate}"
    return np.asarray(audio, dtype=np.float32).reshape(-1)


def list_utterances(root: str):
    items = []
    for trans in sorted(glob.glob(os.path.join(root, "**", "*.trans.txt"), recursive=True)):
        folder = os.path.dirname(trans)
        for line in open(trans, encoding="utf-8"):
            line = line.strip()
            if not line:
                continue
            utt, _, text = line.partition(" ")
            for ext in (".flac", ".wav"):
                if os.path.exists(os.path.join(folder, utt + ext)):
                    items.append((os.path.join(folder, utt + ext), text))
                    break
            else:
                raise SystemExit(f"{trans}: no audio file for utterance {utt}")
    return items


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--root", required=True, help="e.g. ~/.cache/LibriSpeech/test-clean")
    ap.add_argument("--out", default="librispeech.cache")
    ap.add_argument("--batch", type=int, default=16)
    args = ap.parse_args()
    items = list_utterances(os.path.expanduser(args.root))
    if not items:
        raise SystemExit(f"{args.root}: no *.trans.txt found")
    torch.cuda.set_device(0)
    frontend = LogMelFrontend()
    cache = []
    for i in range(0, len(items), args.batch):
        chunk = items[i:i + args.batch]
        wav = np.zeros((len(chunk), N_SAMPLES), dtype=np.float32)   # pad / trim to 30 s like whisper.pad_or_trim
        for j, (path, _) in enumerate(chunk):
            a = read_audio(path)[:N_SAMPLES]
            wav[j, :len(a)] = a
        mel = frontend(torch.from_numpy(wav).cuda()).cpu().numpy()
        cache += [(mel[j], text) for j, (_, text) in enumerate(chunk)]
    with open(args.out, "wb") as f:
        pickle.dump(cache, f)
    print(f"wrote {args.out}: {len(cache)} utterances")
