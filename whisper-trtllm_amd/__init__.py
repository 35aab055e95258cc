"""whisper-trtllm_amd — MI355X-native Whisper encoder-decoder greedy ASR engine.

Mirrors the slice of the `tensorrt_llm` package that examples/whisper uses
(`models.WhisperEncoder/WhisperDecoder`, `builder.Builder`, `network.net_guard`, `runtime.Session/TensorInfo`,
`logger`, `Mapping`, `mpi_rank`) on top of hand-written gfx950 HIP kernels behind a C-ABI
(include/whisper_trtllm_amd.h).  There is no CPU execution path: using a Session without the built HIP
library or without a GPU raises.
"""
from . import _dtypes as trt  # noqa: F401  `trt.float32` spelling used by the reference's run.py
from . import audio, builder, convert, engine_pack, english, generation, layers, models, module, network, runtime, sharding, synthetic, text  # noqa: F401
from .builder import Builder, BuilderConfig  # noqa: F401
from .logger import logger  # noqa: F401
from .models import WhisperDecoder, WhisperEncoder  # noqa: F401
from .network import net_guard  # noqa: F401
from .runtime import DecodeStream, Session, TensorInfo, WhisperDecoderEngine, WhisperEncoderEngine, WhisperPipeline, transcribe_continuous  # noqa: F401

__version__ = "0.1.0"


class Mapping:
    """tensorrt_llm.Mapping(world_size, rank): Whisper always runs Mapping(1, 0) (run.py:235)."""

    def __init__(self, world_size=1, rank=0, tp_size=1):
        self.world_size, self.rank, self.tp_size = world_size, rank, tp_size


def mpi_rank() -> int:
    import os
    return int(os.environ.get("RANK", "0"))
