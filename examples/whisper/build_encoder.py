#!/usr/bin/env python3
"""Build `engine_dir/WhisperEncoder.engine` (+ config.pkl) — same flags and artefacts as the reference's
examples/whisper/build_encoder.py.  `--whisper` is a local HF checkpoint dir or synthetic:<config>[:seed]."""
import argparse
import os
import pickle
import time

from _common import load_checkpoint

import whisper_trtllm_amd as tensorrt_llm
from whisper_trtllm_amd.logger import logger


def serialize_engine(engine, path):
    logger.info(f"Serializing engine to {path}...")
    tik = time.time()
    with open(path, "wb") as f:
        f.write(bytearray(engine))
    logger.info(f"Engine serialized. Total time: {time.strftime('%H:%M:%S', time.gmtime(time.time() - tik))}")


def parse_arguments():
    parser = argparse.ArgumentParser()
    parser.add_argument("--whisper", type=str, default="synthetic:whisper-tiny.en")
    parser.add_argument("--engine_precision", type=str, default="float32")
    parser.add_argument("--log_level", type=str, default="error")
    parser.add_argument("--engine_dir", type=str, default="whisper_outputs")
    return parser.parse_args()


if __name__ == "__main__":
    args = parse_arguments()
    logger.set_level(args.log_level)
    os.makedirs(args.engine_dir, exist_ok=True)
    config, ckpt = load_checkpoint(args.whisper)
    with open(os.path.join(args.engine_dir, "config.pkl"), "wb") as f:
        pickle.dump(config, f)
    engine = tensorrt_llm.convert.build_encoder_engine(config, ckpt, precision=args.engine_precision)
    assert engine is not None, "Failed to build engine"
    serialize_engine(engine, os.path.join(args.engine_dir, "WhisperEncoder.engine"))
