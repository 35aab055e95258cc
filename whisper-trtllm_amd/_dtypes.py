"""Element types of engine tensors.

The reference's scripts name dtypes through TensorRT (`trt.float32`, run.py:22-34); TensorRT does not
exist on ROCm, so this module provides the same spellings as plain objects that map onto the C-ABI's
`wt_dtype` numbering (include/whisper_trtllm_amd.h).
"""
from __future__ import annotations

import numpy as np


class DataType:
    __slots__ = ("name", "code", "np_dtype", "itemsize")

    def __init__(self, name: str, code: int, np_dtype):
        self.name, self.code, self.np_dtype = name, code, np.dtype(np_dtype)
        self.itemsize = self.np_dtype.itemsize

    def __repr__(self):
        return f"DataType.{self.name}"


float32 = DataType("float32", 0, np.float32)
float16 = DataType("float16", 1, np.float16)
int32 = DataType("int32", 2, np.int32)
int8 = DataType("int8", 3, np.int8)
_BY_NAME = {t.name: t for t in (float32, float16, int32, int8)}
_BY_CODE = {t.code: t for t in (float32, float16, int32, int8)}


def str_dtype_to_trt(name: str) -> DataType:
    """Same helper name as tensorrt_llm/_utils.py (str -> engine dtype)."""
    if name not in _BY_NAME:
        raise ValueError(f"unsupported engine dtype {name!r}")
    return _BY_NAME[name]


def from_code(code: int) -> DataType:
    return _BY_CODE[int(code)]


def torch_dtype(dt: DataType):
    import torch
    return {"float32": torch.float32, "float16": torch.float16, "int32": torch.int32, "int8": torch.int8}[dt.name]


def from_torch(tdtype) -> DataType:
    import torch
    table = {torch.float32: float32, torch.float16: float16, torch.int32: int32, torch.int8: int8}
    if tdtype not in table:
        raise ValueError(f"no engine dtype for {tdtype}")
    return table[tdtype]
