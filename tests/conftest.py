import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")

# The library reads its A/B tuning switches (WT_NSPLIT_SELF, WT_NO_FUSED_ARGMAX, ...) only when WT_TUNING=1 (csrc/wt_common.h:
# tuning_env, latched at first use); a few tests exercise non-default paths through them, so the test process opts in.
os.environ.setdefault("WT_TUNING", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: opt-in (WT_RUN_SLOW=1): minutes of host CPU for the oracle at the headline workload's full size")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN_DIR


def load_case(name):
    """Golden case -> (npz dict, config, numpy weights, mel)."""
    import json

    import numpy as np

    import whisper_trtllm_amd  # noqa: F401
    from whisper_trtllm_amd import synthetic

    z = dict(np.load(os.path.join(GOLDEN_DIR, name + ".npz")))
    cfg = synthetic.get_config(str(z["config_name"]))
    cfg.update(json.loads(str(z["overrides_json"])))
    weights = synthetic.make_weights(cfg, int(z["seed"]))
    mel = synthetic.make_mel(cfg, index=int(z["mel_index"]), batch=int(z["batch"]))
    return z, cfg, weights, mel


def sub(a):
    """Same strided subsample as tests/golden/make_golden.py::sub."""
    import numpy as np
    a = np.asarray(a)
    return np.ascontiguousarray(a[..., ::max(1, a.shape[-2] // 24), ::max(1, a.shape[-1] // 32)])


# tiny-long_b2: a full 447-step generation of the reference (self-cache length / position rows up to 447)
GOLDEN_CASES = ["toy-short_b3", "toy-short-eos1_b3", "toy-short-eosall_b3", "toy-wide_b2", "toy_b1", "tiny_b2", "tiny-long_b2"]
