#!/usr/bin/env python3
"""Is the CPU oracle a fair speed proxy for the reference's own CPU path?  (build container only.)

Times the REFERENCE's bundled HuggingFace `generate()` and this repo's `oracle/cpu_ref.py` on the same seeded weights, the same
synthetic log-mel and the same number of greedy steps, same thread count, 2nd pass timed as in run.py:296-315.  The reference
never travels to the GPU box, where `bench.py`'s `cpu_baseline` therefore times the oracle; this script records the ratio
between the two here.  Usage: PYTHONDONTWRITEBYTECODE=1 python tests/golden/time_reference_vs_oracle.py [config] [steps]"""
import contextlib
import io
import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import ROOT, build_hf, import_reference_hf  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "whisper-tiny.en"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    sys.path.insert(0, ROOT)
    import whisper_trtllm_amd  # noqa: F401
    from whisper_trtllm_amd import synthetic
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import cpu_ref
    WhisperConfig, Model, _ = import_reference_hf()
    threads = len(os.sched_getaffinity(0))
    torch.set_num_threads(threads)
    cfg = dict(synthetic.get_config(name))
    cfg["max_length"] = steps + 1
    weights = synthetic.make_weights(cfg, 0)
    mel = torch.from_numpy(synthetic.make_mel(cfg, index=0, batch=1))
    model = build_hf(cfg, weights, WhisperConfig, Model)
    W = cpu_ref.to_torch(weights)

    def run_reference():
        with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):   # the reference prints from its stopping criteria
            return model.generate(mel, max_length=steps + 1)

    def run_oracle():
        with torch.no_grad():
            h = cpu_ref.encoder_forward(W, cfg, mel)
            return cpu_ref.greedy_search(W, cfg, h, max_length=steps + 1)

    out = {}
    for label, fn in (("reference", run_reference), ("oracle", run_oracle)):
        fn()                                    # 1st pass: warm-up, as in run.py:296
        t0 = time.perf_counter()
        ids = fn()
        out[label] = (time.perf_counter() - t0, ids)
    same = torch.equal(torch.as_tensor(out["reference"][1]).long(), torch.as_tensor(out["oracle"][1]).long())
    tr, to = out["reference"][0], out["oracle"][0]
    print(f"{name}, 1 utterance, encoder + {steps} greedy steps, {threads} threads: reference HF generate {tr:.3f} s, oracle {to:.3f} s, "
          f"oracle / reference = {to / tr:.2f}, ids identical: {same}")


if __name__ == "__main__":
    main()
