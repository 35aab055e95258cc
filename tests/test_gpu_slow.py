"""Opt-in GPU checks that need minutes of host CPU for the oracle: marked `gpu` AND `slow`, skipped unless WT_RUN_SLOW=1.

    WT_RUN_SLOW=1 python -m pytest tests/test_gpu_slow.py -m "gpu and slow" -x -q -s

`test_headline_workload_matches_oracle` is the headline benchmark's own workload (whisper-medium.en fp32, batch 8, all 447 decoder
steps = 3576 greedy decisions) against the CPU oracle: every id, every logit of every step (<= 1e-3, the north star's tolerance)
and the encoder output.  The oracle needs ~200 s of 16 host cores for it, which is why the default `-m gpu` run covers medium.en
at 5 steps only (tests/test_gpu_configs.py) and this test is opt-in; the last recorded runs are profiles/r02_headline_parity_vs_oracle.txt and
profiles/r03f_headline_parity_vs_oracle.txt (ids equal, logits within 1.8e-5, minimum top-2 margin of the oracle 1.4e-4)."""
import os
import time

import numpy as np
import pytest
import torch

pytestmark = [pytest.mark.gpu, pytest.mark.slow]


@pytest.mark.skipif(os.environ.get("WT_RUN_SLOW") != "1", reason="opt-in: set WT_RUN_SLOW=1 (about 4 minutes of 16 host cores)")
def test_headline_workload_matches_oracle():
    import cpu_ref
    import whisper_trtllm_amd as w
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU is visible")
    cfg = w.synthetic.get_config("whisper-medium.en")
    weights = w.synthetic.make_weights(cfg, 77)
    B, steps = 8, 447
    mel = w.synthetic.make_mel(cfg, index=300, batch=B)
    enc = w.WhisperEncoderEngine(w.convert.build_encoder_engine(cfg, weights))
    dec = w.WhisperDecoderEngine(w.convert.build_decoder_engine(cfg, weights), cfg)
    hidden = enc(torch.from_numpy(mel).cuda())
    trace = torch.zeros(B, steps, cfg["vocab_size"], dtype=torch.float32, device="cuda")
    ids = dec.generate(hidden, logits_trace=trace).cpu()
    torch.set_num_threads(16)
    W = cpu_ref.to_torch(weights)
    t0 = time.time()
    with torch.no_grad():
        h = cpu_ref.encoder_forward(W, cfg, torch.from_numpy(mel))
        ids_ref, logits_ref = cpu_ref.greedy_search(W, cfg, h, max_length=steps + 1, return_logits=True)
    print(f"oracle: {time.time() - t0:.0f} s", flush=True)
    top2 = torch.topk(logits_ref[:, 1:], 2, dim=-1).values
    margin = float((top2[..., 0] - top2[..., 1]).min())
    enc_err = float((hidden.cpu() - h).abs().max())
    err = (trace.cpu() - logits_ref).abs().amax(dim=(0, 2))
    print(f"min top-2 margin {margin:.3e}; encoder err {enc_err:.3e} (scale {float(h.abs().max()):.2f}); "
          f"max logits err {float(err.max()):.3e} at step {int(err.argmax())}", flush=True)
    assert enc_err < 2e-4
    assert float(err.max()) < 1e-3
    assert float(err.max()) * 4 < margin, "the oracle's own top-2 margin is too thin for an id comparison on this seed"
    assert torch.equal(ids, ids_ref.to(ids.dtype)), (ids != ids_ref).nonzero()[:3].tolist()
