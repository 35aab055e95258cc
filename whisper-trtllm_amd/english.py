"""English text normalisation for the WER harness.

The reference scores transcripts after `whisper.normalizers.EnglishTextNormalizer` (examples/whisper/cal_wer.py:11, 281-284);
the same algorithm ships in the bundled transformers tree (models/whisper/english_normalizer.py:94-595), which is what this
module restates without importing either package: contractions and titles are expanded, bracketed asides and filler words are
dropped, spelled-out numbers become digits ("twenty one dollars and seven cents" -> "$21.07", "one oh one" -> "101"),
punctuation and diacritics go, and an optional British->American spelling table (the checkpoint's normalizer.json) is applied.

Pinned by tests/golden/normalizer.json: 951 inputs (hand-written sentences + seeded word salads over the number vocabulary)
recorded from the reference implementation by tests/golden/make_golden_normalizer.py; tests/test_text.py checks every one.
"""
from __future__ import annotations

import re
import unicodedata
from fractions import Fraction
from typing import Dict, Iterable, List, Mapping, Optional, Tuple, Union

# letters NFKD does not decompose (english_normalizer.py:25-42)
_EXTRA_FOLDS = {"œ": "oe", "Œ": "OE", "ø": "o", "Ø": "O", "æ": "ae", "Æ": "AE", "ß": "ss", "ẞ": "SS", "đ": "d", "Đ": "D",
                "ð": "d", "Ð": "D", "þ": "th", "Þ": "th", "ł": "l", "Ł": "L"}


def strip_symbols(text: str, fold_diacritics: bool = False, keep: str = "") -> str:
    """Marks, symbols and punctuation (Unicode categories M*, S*, P*) become spaces.  With `fold_diacritics` the text is
    decomposed first (NFKD), combining marks vanish and the letters of `_EXTRA_FOLDS` are spelled out; characters in `keep`
    survive (english_normalizer.py:45-72)."""
    out = []
    if not fold_diacritics:
        for ch in unicodedata.normalize("NFKC", text):
            out.append(" " if unicodedata.category(ch)[0] in "MSP" else ch)
        return "".join(out)
    for ch in unicodedata.normalize("NFKD", text):
        if ch in keep:
            out.append(ch)
        elif ch in _EXTRA_FOLDS:
            out.append(_EXTRA_FOLDS[ch])
        else:
            cat = unicodedata.category(ch)
            if cat == "Mn":
                continue
            out.append(" " if cat[0] in "MSP" else ch)
    return "".join(out)


_BRACKETED = re.compile(r"[<\[][^>\]]*[>\]]")
_PARENTHESISED = re.compile(r"\(([^)]+?)\)")
_SPACES = re.compile(r"\s+")


class BasicTextNormalizer:
    """Lower-case, drop bracketed spans, symbols -> spaces, collapse whitespace (english_normalizer.py:75-91; no trimming)."""

    def __init__(self, remove_diacritics: bool = False):
        self.remove_diacritics = remove_diacritics

    def __call__(self, text: str) -> str:
        s = _PARENTHESISED.sub("", _BRACKETED.sub("", text.lower()))
        s = strip_symbols(s, fold_diacritics=self.remove_diacritics).lower()
        return _SPACES.sub(" ", s)


# ---------------------------------------------------------------------------------------------- spelled-out numbers
_UNITS = ["one", "two", "three", "four", "five", "six", "seven", "eight", "nine", "ten", "eleven", "twelve", "thirteen", "fourteen",
          "fifteen", "sixteen", "seventeen", "eighteen", "nineteen"]
_DECADES = ["twenty", "thirty", "forty", "fifty", "sixty", "seventy", "eighty", "ninety"]
_SCALES = ["hundred", "thousand", "million", "billion", "trillion", "quadrillion", "quintillion", "sextillion", "septillion",
           "octillion", "nonillion", "decillion"]
_IRREGULAR_ORDINALS = {"zeroth": 0, "first": 1, "second": 2, "third": 3, "fifth": 5, "twelfth": 12}
_SIGN_WORDS = {"minus": "-", "negative": "-", "plus": "+", "positive": "+"}
_CURRENCY_WORDS = {"pound": "£", "pounds": "£", "euro": "€", "euros": "€", "dollar": "$", "dollars": "$", "cent": "¢", "cents": "¢"}
_NUMERIC = re.compile(r"^\d+(\.\d+)?$")

Value = Union[int, str, None]


def _number_vocabulary() -> Dict[str, Tuple[str, object]]:
    """word -> (kind, payload).  Kinds: zero | unit | unit+ | decade | decade+ | scale | scale+ | sign | currency | percent | glue
    ('+' = carries a plural/ordinal suffix and therefore ends the number)."""
    vocab: Dict[str, Tuple[str, object]] = {}
    for w in ("o", "oh", "zero"):
        vocab[w] = ("zero", 0)
    for i, w in enumerate(_UNITS, start=1):
        vocab[w] = ("unit", i)
        vocab["sixes" if w == "six" else w + "s"] = ("unit+", (i, "s"))
        if i > 3 and i not in (5, 12):
            vocab[w + ("h" if w.endswith("t") else "th")] = ("unit+", (i, "th"))
    for w, i in _IRREGULAR_ORDINALS.items():
        vocab[w] = ("unit+", (i, {1: "st", 2: "nd", 3: "rd"}.get(i, "th")))
    for k, w in enumerate(_DECADES, start=2):
        vocab[w] = ("decade", 10 * k)
        vocab[w.replace("y", "ies")] = ("decade+", (10 * k, "s"))
        vocab[w.replace("y", "ieth")] = ("decade+", (10 * k, "th"))
    for k, w in enumerate(_SCALES):
        mult = 100 if k == 0 else 1000 ** k
        vocab[w] = ("scale", mult)
        vocab[w + "s"] = ("scale+", (mult, "s"))
        vocab[w + "th"] = ("scale+", (mult, "th"))
    for w, sym in _SIGN_WORDS.items():
        vocab[w] = ("sign", sym)
    for w, sym in _CURRENCY_WORDS.items():
        vocab[w] = ("currency", sym)
    vocab["per"] = ("percent", {"cent": "%"})
    vocab["percent"] = ("percent", "%")
    for w in ("and", "double", "triple", "point"):
        vocab[w] = ("glue", w)
    return vocab


class EnglishNumberNormalizer:
    """Spelled-out numbers -> digits over a lower-cased, punctuation-free word stream (english_normalizer.py:94-493):
    commas are gone already, suffixes survive ("1960s", "32nd"), currency words move in front as symbols, "one"/"ones" stay
    words, runs of single digits are read as a nominal number ("one oh one" -> 101)."""

    def __init__(self):
        self.vocab = _number_vocabulary()
        self.symbols = set(_SIGN_WORDS.values()) | set(_CURRENCY_WORDS.values())
        self.decimal_words = {w for w, (kind, _) in self.vocab.items() if kind in ("zero", "unit", "decade")}
        self.scale_words = {w for w, (kind, _) in self.vocab.items() if kind == "scale"}
        self.unit_words = {w for w, (kind, _) in self.vocab.items() if kind == "unit"}
        self.decade_words = {w for w, (kind, _) in self.vocab.items() if kind == "decade"}
        self.zero_words = {w for w, (kind, _) in self.vocab.items() if kind == "zero"}

    # -- the three stages ------------------------------------------------------------------------------------------
    def __call__(self, text: str) -> str:
        words = self._split_for_numbers(text).split()
        return self._tidy(" ".join(self._convert(words)))

    def _split_for_numbers(self, s: str) -> str:
        # "<number> and a half" -> "<number> point five"; anything else keeps its "and a half"
        pieces = re.split(r"\band\s+a\s+half\b", s)
        kept: List[str] = []
        for i, piece in enumerate(pieces):
            if not piece.strip():
                continue
            kept.append(piece)
            if i != len(pieces) - 1:
                last = piece.rsplit(maxsplit=2)[-1]
                kept.append("point five" if (last in self.decimal_words or last in self.scale_words) else "and a half")
        s = " ".join(kept)
        s = re.sub(r"([a-z])([0-9])", r"\1 \2", s)            # letter|digit boundaries get a space ...
        s = re.sub(r"([0-9])([a-z])", r"\1 \2", s)
        return re.sub(r"([0-9])\s+(st|nd|rd|th|s)\b", r"\1\2", s)  # ... except before an ordinal / plural suffix

    @staticmethod
    def _tidy(s: str) -> str:
        def merge_cents(m):
            return f"{m.group(1)}{m.group(2)}.{int(m.group(3)):02d}"

        s = re.sub(r"([€£$])([0-9]+) (?:and )?¢([0-9]{1,2})\b", merge_cents, s)      # "$2 and ¢7" -> "$2.07"
        s = re.sub(r"[€£$]0.([0-9]{1,2})\b", lambda m: f"¢{int(m.group(1))}", s)      # "$0.75" -> "¢75"
        return re.sub(r"\b1(s?)\b", r"one\1", s)                                       # keep "one(s)" readable

    # -- the word-level state machine --------------------------------------------------------------------------------
    def _convert(self, words: List[str]) -> Iterable[str]:
        pending: Value = None          # the number being assembled: int while it is a plain quantity, str once digits are glued
        sign: Optional[str] = None     # symbol to put in front of the next emitted item
        out: List[str] = []

        def emit(item) -> None:
            nonlocal pending, sign
            text = str(item)
            out.append(text if sign is None else sign + text)
            pending, sign = None, None

        def flush() -> None:
            if pending is not None:
                emit(pending)

        def glued(amount: int, previous: Optional[str]) -> Value:
            """`pending` extended by a unit (1..19) that cannot simply be added to it."""
            if isinstance(pending, str) or previous in self.unit_words:
                if previous in self.decade_words and amount < 10:
                    return str(pending)[:-1] + str(amount)      # "twenty" + "one" after digits: overwrite the trailing zero
                return str(pending) + str(amount)
            room = 10 if amount < 10 else 100
            return pending + amount if pending % room == 0 else str(pending) + str(amount)

        def scaled(mult: int):
            """`pending` (a str or the int 0) times a scale word; None when the product is not an integer."""
            try:
                frac = Fraction(pending)
            except ValueError:
                return None
            prod = frac * mult
            return prod.numerator if prod.denominator == 1 else None

        skip_next = False
        for i, word in enumerate(words):
            if skip_next:
                skip_next = False
                continue
            before = words[i - 1] if i > 0 else None
            after = words[i + 1] if i + 1 < len(words) else None
            after_is_digits = after is not None and _NUMERIC.match(after) is not None
            after_is_number_word = after in self.vocab
            signed = word[0] in self.symbols
            bare = word[1:] if signed else word

            if _NUMERIC.match(bare):                                  # digits, perhaps signed, perhaps with a decimal part
                frac = Fraction(bare)
                if pending is not None:
                    if isinstance(pending, str) and pending.endswith("."):
                        pending = str(pending) + str(word)            # "three point" + "14", dotted quads
                        continue
                    emit(pending)
                if signed:
                    sign = word[0]
                pending = frac.numerator if frac.denominator == 1 else bare
                continue

            entry = self.vocab.get(word)
            if entry is None:                                         # an ordinary word ends any number
                flush()
                emit(word)
                continue
            kind, payload = entry

            if kind == "zero":
                pending = str(pending or "") + "0"
            elif kind == "unit":
                pending = payload if pending is None else glued(payload, before)
            elif kind == "unit+":
                amount, suffix = payload
                emit(str(amount) + suffix if pending is None else str(glued(amount, before)) + suffix)
            elif kind == "decade":
                if pending is None:
                    pending = payload
                elif isinstance(pending, str):
                    pending = pending + str(payload)
                else:
                    pending = pending + payload if pending % 100 == 0 else str(pending) + str(payload)
            elif kind == "decade+":
                amount, suffix = payload
                if pending is None:
                    emit(str(amount) + suffix)
                elif isinstance(pending, str):
                    emit(pending + str(amount) + suffix)
                else:
                    emit((str(pending + amount) if pending % 100 == 0 else str(pending) + str(amount)) + suffix)
            elif kind == "scale":
                if pending is None:
                    pending = payload
                elif isinstance(pending, str) or pending == 0:
                    product = scaled(payload)
                    if product is None:
                        emit(pending)
                        pending = payload
                    else:
                        pending = product
                else:                                                 # "two thousand three" + "hundred": scale the last group only
                    pending = pending // 1000 * 1000 + pending % 1000 * payload
            elif kind == "scale+":
                mult, suffix = payload
                if pending is None:
                    emit(str(mult) + suffix)
                elif isinstance(pending, str):
                    product = scaled(mult)
                    if product is None:
                        emit(pending)
                        emit(str(mult) + suffix)
                    else:
                        emit(str(product) + suffix)
                else:
                    emit(str(pending // 1000 * 1000 + pending % 1000 * mult) + suffix)
            elif kind == "sign":                                      # "minus five" -> "-5"; "minus" alone stays a word
                flush()
                if after_is_number_word or after_is_digits:
                    sign = payload
                else:
                    emit(word)
            elif kind == "currency":                                  # "five dollars" -> "$5"; "dollars" alone stays a word
                if pending is not None:
                    sign = payload
                    emit(pending)
                else:
                    emit(word)
            elif kind == "percent":
                if pending is None:
                    emit(word)
                elif isinstance(payload, dict):                       # "per" needs "cent" behind it
                    if after in payload:
                        emit(str(pending) + payload[after])
                        skip_next = True
                    else:
                        emit(pending)
                        emit(word)
                else:
                    emit(str(pending) + payload)
            else:                                                     # glue words: and / double / triple / point
                if not after_is_number_word and not after_is_digits:
                    flush()
                    emit(word)
                elif word == "and":
                    if before not in self.scale_words:                # "hundred and five": the "and" is swallowed
                        flush()
                        emit(word)
                elif word in ("double", "triple"):
                    if after in self.unit_words or after in self.zero_words:
                        digit = self.vocab[after][1]
                        pending = str(pending or "") + str(digit) * (2 if word == "double" else 3)
                        skip_next = True
                    else:
                        flush()
                        emit(word)
                elif after in self.decimal_words or after_is_digits:   # "point" in front of digits
                    pending = str(pending or "") + "."
        flush()
        return out


# ------------------------------------------------------------------------------------------------- the full pipeline
_FILLERS = re.compile(r"\b(hmm|mm|mhm|mmm|uh|um)\b")
# order matters: specific contractions first, general suffixes last (english_normalizer.py:514-575)
_REWRITES: List[Tuple[str, str]] = [
    (r"\bwon't\b", "will not"), (r"\bcan't\b", "can not"), (r"\blet's\b", "let us"), (r"\bain't\b", "aint"), (r"\by'all\b", "you all"),
    (r"\bwanna\b", "want to"), (r"\bgotta\b", "got to"), (r"\bgonna\b", "going to"), (r"\bi'ma\b", "i am going to"),
    (r"\bimma\b", "i am going to"), (r"\bwoulda\b", "would have"), (r"\bcoulda\b", "could have"), (r"\bshoulda\b", "should have"),
    (r"\bma'am\b", "madam"),
    (r"\bmr\b", "mister "), (r"\bmrs\b", "missus "), (r"\bst\b", "saint "), (r"\bdr\b", "doctor "), (r"\bprof\b", "professor "),
    (r"\bcapt\b", "captain "), (r"\bgov\b", "governor "), (r"\bald\b", "alderman "), (r"\bgen\b", "general "), (r"\bsen\b", "senator "),
    (r"\brep\b", "representative "), (r"\bpres\b", "president "), (r"\brev\b", "reverend "), (r"\bhon\b", "honorable "),
    (r"\basst\b", "assistant "), (r"\bassoc\b", "associate "), (r"\blt\b", "lieutenant "), (r"\bcol\b", "colonel "), (r"\bjr\b", "junior "),
    (r"\bsr\b", "senior "), (r"\besq\b", "esquire "),
    (r"'d been\b", " had been"), (r"'s been\b", " has been"), (r"'d gone\b", " had gone"), (r"'s gone\b", " has gone"),
    (r"'d done\b", " had done"), (r"'s got\b", " has got"),
    (r"n't\b", " not"), (r"'re\b", " are"), (r"'s\b", " is"), (r"'d\b", " would"), (r"'ll\b", " will"), (r"'t\b", " not"),
    (r"'ve\b", " have"), (r"'m\b", " am"),
]
_REWRITES_COMPILED = [(re.compile(p), r) for p, r in _REWRITES]


class EnglishTextNormalizer:
    """`EnglishTextNormalizer(spelling)(text)` == the reference's normaliser (english_normalizer.py:508-595).  `spelling` is the
    British->American word table of the checkpoint's normalizer.json (None / {} = leave spellings alone)."""

    def __init__(self, english_spelling_mapping: Optional[Mapping[str, str]] = None):
        self.spelling = dict(english_spelling_mapping or {})
        self.numbers = EnglishNumberNormalizer()

    def __call__(self, text: str) -> str:
        s = _PARENTHESISED.sub("", _BRACKETED.sub("", text.lower()))
        s = _FILLERS.sub("", s)
        s = re.sub(r"\s+'", "'", s)                              # "it 's" -> "it's"
        for pattern, replacement in _REWRITES_COMPILED:
            s = pattern.sub(replacement, s)
        s = re.sub(r"(\d),(\d)", r"\1\2", s)                     # thousands separators
        s = re.sub(r"\.([^0-9]|$)", r" \1", s)                   # full stops, but not decimal points
        s = strip_symbols(s, fold_diacritics=True, keep=".%$¢€£")
        s = self.numbers(s)
        s = " ".join(self.spelling.get(w, w) for w in s.split())
        s = re.sub(r"[.$¢€£]([^0-9])", r" \1", s)                # symbols that did not end up next to a number
        s = re.sub(r"([^0-9])%", r"\1 ", s)
        return _SPACES.sub(" ", s)
