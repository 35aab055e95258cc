#!/usr/bin/env python3
"""Golden vectors for ids -> text from the REFERENCE's bundled WhisperTokenizer (build container only).
A toy byte-level BPE vocabulary (256 byte symbols + a few merges + Whisper-style special tokens) is written to a temp dir,
the reference tokenizer decodes seeded id sequences, and (vocab, added tokens, ids, texts) go to tests/golden/text.json."""
import json
import os
import random
import sys
import tempfile
import types

HERE = os.path.dirname(os.path.abspath(__file__))
REF_SRC = "/root/reference/transformers/src"


def main():
    stub = types.ModuleType("transformers.dependency_versions_check")
    stub.dep_version_check = lambda *a, **k: None
    sys.modules["transformers.dependency_versions_check"] = stub
    sys.path.insert(0, REF_SRC)
    import transformers
    assert transformers.__version__ == "4.33.0.dev0" and transformers.__file__.startswith(REF_SRC)
    from transformers.models.whisper.tokenization_whisper import WhisperTokenizer, bytes_to_unicode

    b2u = bytes_to_unicode()
    symbols = [b2u[b] for b in range(256)]
    merges = [("Ġ", "t"), ("h", "e"), ("Ġt", "he"), ("Ã", "©"), ("Ġ", "w"), ("o", "r"), ("Ġw", "or"), ("l", "d"), ("Ġwor", "ld")]
    vocab = {s: i for i, s in enumerate(symbols)}
    for a, b in merges:
        vocab[a + b] = len(vocab)
    vocab["<|endoftext|>"] = len(vocab)
    tmp = tempfile.mkdtemp()
    json.dump(vocab, open(os.path.join(tmp, "vocab.json"), "w", encoding="utf-8"), ensure_ascii=False)
    open(os.path.join(tmp, "merges.txt"), "w", encoding="utf-8").write("#version: 0.2\n" + "\n".join(f"{a} {b}" for a, b in merges) + "\n")
    tok = WhisperTokenizer(os.path.join(tmp, "vocab.json"), os.path.join(tmp, "merges.txt"))
    extra = ["<|startoftranscript|>", "<|notimestamps|>", "<|startofprev|>"]
    tok.add_special_tokens({"additional_special_tokens": extra})
    added = {t: tok.convert_tokens_to_ids(t) for t in extra}
    eot = vocab["<|endoftext|>"]
    rng = random.Random(0)
    seqs = [
        [added["<|startoftranscript|>"], added["<|notimestamps|>"]] + tok.encode(" the world", add_special_tokens=False) + [eot],
        tok.encode(" café the wörld!", add_special_tokens=False),
        [added["<|startoftranscript|>"]] + [rng.randrange(0, 256 + len(merges)) for _ in range(40)] + [eot, eot],   # invalid utf-8 runs
        [],
        [eot],
    ] + [[rng.randrange(0, len(vocab)) for _ in range(rng.randrange(1, 30))] for _ in range(20)]
    out = {"vocab": vocab, "added_tokens": added, "cases": []}
    for ids in seqs:
        out["cases"].append({"ids": ids, "skip": tok.decode(ids, skip_special_tokens=True), "keep": tok.decode(ids, skip_special_tokens=False)})
    json.dump(out, open(os.path.join(HERE, "text.json"), "w", encoding="utf-8"), ensure_ascii=False, indent=0)
    print("wrote text.json with", len(seqs), "cases; example:", repr(out["cases"][0]["skip"]), repr(out["cases"][1]["keep"]))


if __name__ == "__main__":
    main()
