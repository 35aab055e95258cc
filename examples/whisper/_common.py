"""Shared helpers of the example scripts: checkpoint loading (`--whisper`) and repo path setup."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def load_checkpoint(spec: str):
    """`--whisper` -> (config dict, state_dict of numpy arrays keyed by HF names).

    * `synthetic:<config>[:<seed>]` — seeded random-init weights of the real architecture (no checkpoints or
      network exist on the build/GPU boxes); e.g. synthetic:whisper-tiny.en:0
    * a local directory cloned from the HF hub, as in the reference (README.md:29-31): config.json +
      model.safetensors (or pytorch_model.bin, loaded with weights_only=True)."""
    import numpy as np
    import whisper_trtllm_amd as wt
    if spec.startswith("synthetic:"):
        parts = spec.split(":")
        cfg = wt.synthetic.get_config(parts[1])
        return cfg, wt.synthetic.make_weights(cfg, int(parts[2]) if len(parts) > 2 else 0)
    if not os.path.isdir(spec):
        raise SystemExit(f"--whisper {spec!r}: not a local checkpoint directory and not 'synthetic:<config>[:seed]' "
                         f"(known configs: {sorted(wt.synthetic.CONFIGS)})")
    cfg = json.load(open(os.path.join(spec, "config.json")))
    gen_path = os.path.join(spec, "generation_config.json")
    if os.path.exists(gen_path):
        for k, v in json.load(open(gen_path)).items():
            cfg.setdefault(k, v)
    cfg.setdefault("forced_bos_token_id", None)
    st_path, bin_path = os.path.join(spec, "model.safetensors"), os.path.join(spec, "pytorch_model.bin")
    if os.path.exists(st_path):
        from safetensors.numpy import load_file
        sd = load_file(st_path)
    elif os.path.exists(bin_path):
        import torch
        sd = {k: v.float().numpy() for k, v in torch.load(bin_path, map_location="cpu", weights_only=True).items()}
    else:
        raise SystemExit(f"{spec}: no model.safetensors / pytorch_model.bin")
    sd = {k: np.asarray(v, dtype=np.float32) for k, v in sd.items()}
    if "proj_out.weight" not in sd:
        sd["proj_out.weight"] = sd["model.decoder.embed_tokens.weight"]   # tied (modeling_whisper.py:1335)
    return cfg, sd
