#!/usr/bin/env python3
"""Build `engine_dir/WhisperDecoder.engine` — same flags and artefacts as the reference's
examples/whisper/build_decoder.py."""
import os

from _common import load_checkpoint
from build_encoder import parse_arguments, serialize_engine

import whisper_trtllm_amd as tensorrt_llm
from whisper_trtllm_amd.logger import logger

if __name__ == "__main__":
    args = parse_arguments()
    logger.set_level(args.log_level)
    os.makedirs(args.engine_dir, exist_ok=True)
    config, ckpt = load_checkpoint(args.whisper)
    engine = tensorrt_llm.convert.build_decoder_engine(config, ckpt, precision=args.engine_precision)
    assert engine is not None, "Failed to build engine"
    serialize_engine(engine, os.path.join(args.engine_dir, "WhisperDecoder.engine"))
