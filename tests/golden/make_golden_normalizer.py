#!/usr/bin/env python3
"""Golden vectors for the English text normaliser used by the WER harness, recorded from the REFERENCE's
bundled implementation (transformers/models/whisper/english_normalizer.py, the same algorithm as `whisper.normalizers` that the
reference's cal_wer.py:11,281 imports) in the build container only.  Inputs are hand-written sentences plus seeded random word
salads over the number vocabulary, so every branch of the number state machine is exercised.  Output: tests/golden/normalizer.json
(inputs, expected outputs, and the toy spelling table that stands in for the checkpoint's normalizer.json, which is not in the tree)."""
import json
import os
import random
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
REF_SRC = "/root/reference/transformers/src"

HAND = [
    "Mr. Brown won't pay $20 million, he'd rather pay twenty-one dollars and seven cents.",
    "It's one hundred and twenty three thousand four hundred fifty six people.",
    "The 1960s were followed by the nineteen seventies; she came in 274th, he came thirty second.",
    "I'ma go to St. Louis with Dr. Smith [laughter] (inaudible) uh, hmm, maybe.",
    "Call me at one oh one, or double oh seven, or triple five one two.",
    "Three and a half million, two point five, minus ten, plus seven, negative three point one four.",
    "Fifty percent of them, 20% of us, five per cent overall, per se.",
    "One of the ones who paid £5 and 7 cents; €3 and ¢50; $0.75; two pounds ten.",
    "She said y'all gotta wanna gonna coulda shoulda woulda, ma'am. Let's go, can't wait, ain't it?",
    "Prof. Lt. Col. Jr. Sr. Esq. Gen. Sen. Rep. Pres. Rev. Hon. Asst. Assoc. Capt. Gov. Ald. Mrs.",
    "They've been here; he's gone; she'd done it; it's got to stop; I'm sure we'll see you're right.",
    "naïve café Ærøskøbing straße Łódź þing ðæt œuvre",
    "1,234,567 and 3.14159 and 10.0.0.1 and 12th and 3rd and 21st and 2nd.",
    "zero point zero five, oh point five, point nine, nine point, point.",
    "a hundred, a thousand and one, two thousand twenty three, twenty twenty three, nineteen ninety nine",
    "first second third fourth fifth sixth seventh eighth ninth tenth eleventh twelfth twentieth thirtieth hundredth thousandth millionth",
    "sixes and sevens, twenties, thirties, hundreds of thousands, millions of dollars, two billions",
    "one, ones, 1, 1s, one's, one one, one one one",
    "seven hundred billion trillion, five million thousand, half a million, two hundred point five thousand",
    "and and a half and a half, two and a half, million and a half, ten and a half dollars",
    "", "   ", "...", "$", "%", "5 %", "$ 5", "- 5", "+5", "-5", "5-", "5th", "5 th", "mr", "o", "oh", "oh oh", "o o o",
    "THE QUICK BROWN FOX; the quick brown fox!", "hello <noise> world [music] (applause) done",
    "won't can't couldn't shouldn't mustn't isn't aren't wasn't weren't hasn't haven't hadn't doesn't don't didn't",
    "twenty-first century, thirty-second note, forty five rpm, ninety nine point nine percent",
    "eleven hundred, twelve thousand, thirteen million, nineteen eighty four, twenty ten, ten twenty",
    "three dollars fifty, three fifty dollars, fifty cents, a dollar fifty, 3 dollars and 50 cents",
    "one point five billion dollars, $1.5 billion, €2 million, £3.50, ¢99, 99¢",
    "triple a, double b, double nine nine, triple zero, double o seven",
    "minus, plus, positive vibes, negative space, minus one, plus two, positive three, negative four",
    "hundred, thousand, million; hundreds, thousands; hundredth, thousandths",
    "two thirds, three quarters, one half, a half, half",
    "it's 5 o'clock, 5pm, 5 p.m., 10am, 3x, x3, b2b, 4ever, mp3, 24/7, 9/11, 50-50, 1-800-555-1234",
]

NUM_POOL = ["zero", "o", "oh", "one", "two", "three", "four", "five", "six", "seven", "eight", "nine", "ten", "eleven", "twelve",
            "thirteen", "fifteen", "nineteen", "twenty", "thirty", "forty", "fifty", "ninety", "hundred", "thousand", "million",
            "billion", "trillion", "first", "second", "third", "fifth", "ninth", "twelfth", "twentieth", "fortieth", "hundredth",
            "thousandth", "millionth", "ones", "twos", "sixes", "tens", "twenties", "fifties", "hundreds", "thousands", "millions",
            "and", "double", "triple", "point", "minus", "negative", "plus", "positive", "pound", "pounds", "euro", "euros", "dollar",
            "dollars", "cent", "cents", "per", "percent", "a", "half", "1", "2", "7", "10", "12", "20", "100", "1000", "3.5", "0.25", "1960s",
            "21st", "42nd", "3rd", "5th", "$5", "$20", "€7", "£3", "¢50", "+4", "-9", "20%", "7%", "1,000", "2,500,000", "the", "of", "cats",
            "at", "about", "in"]


def main():
    stub = types.ModuleType("transformers.dependency_versions_check")
    stub.dep_version_check = lambda *a, **k: None
    sys.modules["transformers.dependency_versions_check"] = stub
    sys.path.insert(0, REF_SRC)
    import transformers
    assert transformers.__version__ == "4.33.0.dev0" and transformers.__file__.startswith(REF_SRC)
    from transformers.models.whisper.english_normalizer import BasicTextNormalizer, EnglishNumberNormalizer, EnglishTextNormalizer

    mapping = {"colour": "color", "centre": "center", "organise": "organize", "travelling": "traveling", "grey": "gray",
               "theatre": "theater", "programme": "program", "cheque": "check"}
    rng = random.Random(1234)
    inputs = list(HAND)
    inputs += ["The colour of the theatre programme at the centre was grey; travelling costs a cheque of fifty pounds."]
    for _ in range(900):
        n = rng.randint(1, 9)
        inputs.append(" ".join(rng.choice(NUM_POOL) for _ in range(n)))
    full, full_nomap, number, basic, basic_diac = (EnglishTextNormalizer(mapping), EnglishTextNormalizer({}), EnglishNumberNormalizer(),
                                                   BasicTextNormalizer(), BasicTextNormalizer(remove_diacritics=True))
    cases = []
    for s in inputs:
        row = {"input": s}
        for key, fn in (("english", full), ("english_no_spelling", full_nomap), ("number", number), ("basic", basic), ("basic_diacritics", basic_diac)):
            try:
                row[key] = fn(s.lower() if key == "number" else s)
            except Exception as exc:  # the reference raises on a few degenerate inputs: record the exception type
                row[key] = {"raises": type(exc).__name__}
        cases.append(row)
    out = {"source": "transformers 4.33.0.dev0 english_normalizer.py (reference tree)", "spelling_mapping": mapping, "cases": cases}
    path = os.path.join(HERE, "normalizer.json")
    json.dump(out, open(path, "w", encoding="utf-8"), ensure_ascii=False, indent=0)
    n_raise = sum(1 for c in cases for k in c if isinstance(c[k], dict))
    print(f"wrote {path}: {len(cases)} inputs, {n_raise} recorded exceptions, {os.path.getsize(path)} bytes")


if __name__ == "__main__":
    main()
