"""Serialized engine = versioned weight pack (header + tensor table + raw fp32 tensors).

The reference's "engine" is a TensorRT plan (`builder.build_engine` -> bytes written to
`engine_dir/WhisperEncoder.engine`, build_encoder.py:105-109).  Ours keeps the same file names and the
same bytes-in/bytes-out handling, but the payload is this pack; the layout is mirrored by
`csrc/wt_common.h` (BlobHeader / BlobTensor) which `wt_engine_open` parses.
"""
from __future__ import annotations

import struct
from typing import Dict, Tuple

import numpy as np

MAGIC = b"WTENGINE"
VERSION = 2   # 2: decoder packs carry encoder_attn.q_fold.* instead of q_proj / encoder_attn_layer_norm
KIND_ENCODER, KIND_DECODER = 1, 2
_HDR = struct.Struct("<8sIIII24iQQQ")   # 144 bytes
_TEN = struct.Struct("<96sII4qQQ")      # 152 bytes
CFG_KEYS = ("d_model", "n_heads", "n_layers", "ffn_dim", "n_mels", "max_source_positions",
            "max_target_positions", "vocab_size", "tied_proj_out")
_ALIGN = 256


def pack(kind: int, precision_code: int, cfg: Dict[str, int], tensors: Dict[str, np.ndarray]) -> bytes:
    names = list(tensors)
    table_off = _HDR.size
    data_off = (table_off + _TEN.size * len(names) + _ALIGN - 1) // _ALIGN * _ALIGN
    entries, chunks, off = [], [], data_off
    for name in names:
        a = np.ascontiguousarray(tensors[name])
        if a.dtype not in (np.float32, np.float16):
            a = a.astype(np.float32)
        if a.ndim > 4:
            raise ValueError(f"{name}: rank {a.ndim} > 4")
        if len(name.encode()) >= 96:
            raise ValueError(f"tensor name too long: {name}")
        shape = list(a.shape) + [0] * (4 - a.ndim)
        entries.append(_TEN.pack(name.encode(), 0 if a.dtype == np.float32 else 1, a.ndim, *shape, off, a.nbytes))
        pad = (-a.nbytes) % _ALIGN
        chunks.append(a.tobytes() + b"\0" * pad)
        off += a.nbytes + pad
    cfg_arr = [int(cfg.get(k, 0)) for k in CFG_KEYS] + [0] * (24 - len(CFG_KEYS))
    hdr = _HDR.pack(MAGIC, VERSION, kind, precision_code, len(names), *cfg_arr, table_off, data_off, off)
    head = hdr + b"".join(entries)
    return head + b"\0" * (data_off - len(head)) + b"".join(chunks)


def unpack(blob: bytes) -> Tuple[dict, Dict[str, np.ndarray]]:
    """Host-side parser (tests / tooling).  The device path is `wt_engine_open`."""
    if len(blob) < _HDR.size:
        raise ValueError("engine blob too small")
    f = _HDR.unpack_from(blob, 0)
    magic, version, kind, precision, n = f[0], f[1], f[2], f[3], f[4]
    cfg_vals, (table_off, data_off, total) = f[5:29], f[29:32]
    if magic != MAGIC or version != VERSION:
        raise ValueError("not a whisper-trtllm_amd engine blob")
    if total != len(blob):
        raise ValueError("engine blob is truncated")
    info = {"kind": kind, "precision": precision, **dict(zip(CFG_KEYS, cfg_vals))}
    tensors = {}
    for i in range(n):
        name, dt, ndim, s0, s1, s2, s3, off, nb = _TEN.unpack_from(blob, table_off + i * _TEN.size)
        shape = (s0, s1, s2, s3)[:ndim]
        npdt = np.float32 if dt == 0 else np.float16
        tensors[name.rstrip(b"\0").decode()] = np.frombuffer(blob, npdt, nb // np.dtype(npdt).itemsize, off).reshape(shape)
    return info, tensors
