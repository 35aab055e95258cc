"""CPU: ids -> text against golden vectors from the reference's bundled WhisperTokenizer (tests/golden/make_golden_text.py),
and the pooled word error rate against hand-checked cases and edit-distance properties."""
import json
import os
import random

import pytest

import whisper_trtllm_amd  # noqa: F401
from conftest import GOLDEN_DIR
from whisper_trtllm_amd.text import WhisperTokenDecoder, basic_normalize, word_error_rate


@pytest.fixture(scope="module")
def gold():
    return json.load(open(os.path.join(GOLDEN_DIR, "text.json"), encoding="utf-8"))


def test_token_decode_matches_reference(gold):
    dec = WhisperTokenDecoder(gold["vocab"], gold["added_tokens"])
    assert len(gold["cases"]) == 25
    for c in gold["cases"]:
        assert dec.decode(c["ids"], skip_special_tokens=True) == c["skip"], c["ids"]
        assert dec.decode(c["ids"], skip_special_tokens=False) == c["keep"], c["ids"]
    assert dec.batch_decode([c["ids"] for c in gold["cases"][:3]]) == [c["skip"] for c in gold["cases"][:3]]


def test_token_decoder_from_dir(tmp_path, gold):
    json.dump(gold["vocab"], open(tmp_path / "vocab.json", "w", encoding="utf-8"), ensure_ascii=False)
    json.dump(gold["added_tokens"], open(tmp_path / "added_tokens.json", "w", encoding="utf-8"))
    dec = WhisperTokenDecoder.from_dir(str(tmp_path))
    assert dec.decode(gold["cases"][0]["ids"]) == " the world"


def test_word_error_rate_known_cases():
    assert word_error_rate(["the cat sat"], ["the cat sat"]) == 0.0
    assert word_error_rate(["the cat sat"], ["the cat"]) == pytest.approx(1 / 3)          # one deletion
    assert word_error_rate(["the cat sat"], ["the bat sat down"]) == pytest.approx(2 / 3)  # substitution + insertion
    assert word_error_rate(["a b c d", "e f"], ["a b c d", "x y z"]) == pytest.approx(3 / 6)  # pooled over utterances
    assert word_error_rate(["a"], [""]) == 1.0
    with pytest.raises(ValueError):
        word_error_rate(["a"], ["a", "b"])
    with pytest.raises(ValueError):
        word_error_rate([""], ["a"])


def test_word_error_rate_properties():
    rng = random.Random(1)
    words = "a b c d e f g".split()
    for _ in range(50):
        ref = [rng.choice(words) for _ in range(rng.randrange(1, 12))]
        hyp = [rng.choice(words) for _ in range(rng.randrange(0, 12))]
        wer = word_error_rate([" ".join(ref)], [" ".join(hyp)])
        assert abs(len(ref) - len(hyp)) / len(ref) <= wer <= max(len(ref), len(hyp)) / len(ref)
        assert word_error_rate([" ".join(ref)], [" ".join(ref)]) == 0.0


def test_basic_normalize():
    assert basic_normalize("  Hello, [noise] WORLD!  (laughs) It's 5 o'clock. ") == "hello world it s 5 o clock"


def _normalizer_goldens():
    import json
    import os
    path = os.path.join(os.path.dirname(__file__), "golden", "normalizer.json")
    return json.load(open(path, encoding="utf-8"))


def test_english_normalizer_matches_reference_goldens():
    """Every recorded input of tests/golden/normalizer.json (951: hand-written sentences + seeded word salads over the number
    vocabulary) through the five normaliser entry points equals what the reference's english_normalizer.py produced."""
    from whisper_trtllm_amd.english import BasicTextNormalizer, EnglishNumberNormalizer, EnglishTextNormalizer
    g = _normalizer_goldens()
    fns = {"english": EnglishTextNormalizer(g["spelling_mapping"]), "english_no_spelling": EnglishTextNormalizer(),
           "number": EnglishNumberNormalizer(), "basic": BasicTextNormalizer(), "basic_diacritics": BasicTextNormalizer(remove_diacritics=True)}
    assert len(g["cases"]) >= 900
    bad = []
    for case in g["cases"]:
        for key, fn in fns.items():
            got = fn(case["input"].lower() if key == "number" else case["input"])
            if got != case[key]:
                bad.append((key, case["input"], case[key], got))
    assert not bad, f"{len(bad)} mismatches, first: {bad[:3]}"


def test_english_normalizer_examples():
    from whisper_trtllm_amd.english import EnglishTextNormalizer
    n = EnglishTextNormalizer({"colour": "color"})
    assert n("Mr. Brown won't pay $20 million for the colour.") == "mister brown will not pay $20000000 for the color"
    assert n("twenty one dollars and seven cents") == "$21.07"
    assert n("one oh one, double oh seven") == "101007"     # the comma is gone before the digits are read: one nominal number
    assert n("She came in 274th [laughs] (really), uh, three and a half percent.") == "she came in 274th 3.5%"
