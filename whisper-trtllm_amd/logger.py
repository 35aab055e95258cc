"""Severity-filtered logger with the reference's surface (tensorrt_llm/logger.py:30-129): `set_level`, info/error/..."""
from __future__ import annotations

import sys

_LEVELS = {"internal_error": 0, "error": 1, "warning": 2, "info": 3, "verbose": 4}


class Logger:
    def __init__(self):
        self._level = _LEVELS["error"]

    def set_level(self, level: str):
        if level not in _LEVELS:
            raise ValueError(f"log level must be one of {sorted(_LEVELS)}")
        self._level = _LEVELS[level]

    def _emit(self, level: str, msg: str):
        if _LEVELS[level] <= self._level:
            print(f"[whisper-trtllm_amd][{level.upper()}] {msg}", file=sys.stderr)

    def error(self, msg): self._emit("error", msg)
    def warning(self, msg): self._emit("warning", msg)
    def info(self, msg): self._emit("info", msg)
    def verbose(self, msg): self._emit("verbose", msg)
    debug = verbose


logger = Logger()


def set_level(level: str):
    logger.set_level(level)
