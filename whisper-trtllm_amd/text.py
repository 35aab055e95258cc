"""Text side of the evaluation loop: token ids -> text without `transformers`, and WER.

Reference: `hf_processor.batch_decode(predicted_ids, skip_special_tokens=True)` (examples/whisper/run.py:287) ==
WhisperTokenizer._decode / convert_tokens_to_string (transformers/models/whisper/tokenization_whisper.py:562-656), and
`jiwer.wer(references, hypotheses)` after a text normaliser (examples/whisper/cal_wer.py:279-287)."""
from __future__ import annotations

import json
import os
import re
import unicodedata
from typing import Dict, Iterable, List, Optional, Sequence


def _byte_decoder() -> Dict[str, int]:
    """Inverse of GPT-2's printable-byte alphabet: the 188 printable latin-1 bytes map to themselves, the other 68
    bytes to code points 256, 257, ... in increasing byte order."""
    printable = set(range(0x21, 0x7F)) | set(range(0xA1, 0xAD)) | set(range(0xAE, 0x100))
    table, nxt = {}, 256
    for b in range(256):
        if b in printable:
            table[chr(b)] = b
        else:
            table[chr(nxt)] = b
            nxt += 1
    return table


class WhisperTokenDecoder:
    """ids -> text for a byte-level BPE vocabulary (vocab.json: token string -> id)."""

    def __init__(self, vocab: Dict[str, int], added_tokens: Optional[Dict[str, int]] = None,
                 special_ids: Optional[Iterable[int]] = None, errors: str = "replace"):
        self.id_to_token = {i: t for t, i in vocab.items()}
        self.added = dict(added_tokens or {})
        for t, i in self.added.items():
            self.id_to_token[i] = t
        if special_ids is None:  # Whisper: <|endoftext|> and every added <|...|> token are special
            special_ids = [i for t, i in list(vocab.items()) + list(self.added.items()) if t.startswith("<|") and t.endswith("|>")]
        self.special_ids = set(special_ids)
        self.byte_decoder = _byte_decoder()
        self.errors = errors

    @classmethod
    def from_dir(cls, path: str) -> "WhisperTokenDecoder":
        vocab = json.load(open(os.path.join(path, "vocab.json"), encoding="utf-8"))
        added_path = os.path.join(path, "added_tokens.json")
        added = json.load(open(added_path, encoding="utf-8")) if os.path.exists(added_path) else {}
        return cls(vocab, added)

    def _bytes_to_text(self, tokens: List[str]) -> str:
        return bytearray(self.byte_decoder[c] for c in "".join(tokens)).decode("utf-8", errors=self.errors)

    def decode(self, ids: Sequence[int], skip_special_tokens: bool = True) -> str:
        parts, run = [], []
        for i in ids:
            i = int(i)
            if skip_special_tokens and i in self.special_ids:
                continue
            tok = self.id_to_token[i]
            if tok in self.added:          # added tokens are literal text, not byte-level symbols
                if run:
                    parts.append(self._bytes_to_text(run))
                    run = []
                parts.append(tok)
            else:
                run.append(tok)
        if run:
            parts.append(self._bytes_to_text(run))
        return "".join(parts)

    def batch_decode(self, batch: Iterable[Sequence[int]], skip_special_tokens: bool = True) -> List[str]:
        return [self.decode(ids, skip_special_tokens) for ids in batch]


def basic_normalize(text: str) -> str:
    """Lower-case, drop bracketed/parenthesised spans, turn symbols and punctuation into spaces, collapse whitespace
    (the behaviour of the bundled BasicTextNormalizer, english_normalizer.py:75-91, plus a final strip; the full English
    normaliser with contraction, number and spelling rules is `whisper_trtllm_amd.english.EnglishTextNormalizer`)."""
    s = text.lower()
    s = re.sub(r"[<\[][^>\]]*[>\]]", "", s)
    s = re.sub(r"\(([^)]+?)\)", "", s)
    s = "".join(" " if unicodedata.category(c)[0] in "MSP" else c for c in unicodedata.normalize("NFKC", s))
    return re.sub(r"\s+", " ", s).strip()


def _edit_distance(ref: Sequence[str], hyp: Sequence[str]) -> int:
    prev = list(range(len(hyp) + 1))
    for i, r in enumerate(ref, 1):
        cur = [i] + [0] * len(hyp)
        for j, h in enumerate(hyp, 1):
            cur[j] = min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (r != h))
        prev = cur
    return prev[-1]


def word_error_rate(references: Sequence[str], hypotheses: Sequence[str]) -> float:
    """(substitutions + deletions + insertions) / reference words, pooled over all utterances (jiwer.wer semantics)."""
    if len(references) != len(hypotheses):
        raise ValueError("references and hypotheses differ in length")
    errors = words = 0
    for r, h in zip(references, hypotheses):
        rw, hw = r.split(), h.split()
        errors += _edit_distance(rw, hw)
        words += len(rw)
    if words == 0:
        raise ValueError("no reference words")
    return errors / words
