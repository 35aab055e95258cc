"""ctypes binding of the C-ABI (include/whisper_trtllm_amd.h).  Fails loudly: there is NO CPU fallback.

The reference opens its plugin library the same way (tensorrt_llm/plugin/plugin.py:10-22,
ctypes.CDLL(..., RTLD_GLOBAL) + `initLibNvInferPlugins`).
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, Structure, c_char, c_char_p, c_float, c_int, c_int32, c_int64, c_size_t, c_void_p

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "lib", "libwhisper_trtllm_amd.so")
WT_NAME_LEN, WT_MAX_DIMS = 48, 6
ABI_VERSION = 3

EXPORTS = [
    "wt_engine_open", "wt_engine_clone", "wt_engine_close", "wt_engine_get_info", "wt_engine_gemm_mode", "wt_engine_infer_shapes", "wt_engine_run",
    "wt_encoder_forward", "wt_decoder_begin", "wt_decoder_steps", "wt_decoder_poll", "wt_decoder_run", "wt_decoder_read_ids",
    "wt_decoder_greedy", "wt_decoder_stream_begin", "wt_decoder_stream_submit", "wt_decoder_stream_run", "wt_decoder_stream_collect",
    "wt_engine_set_profiling", "wt_engine_get_timer", "wt_decoder_time_cross_attention", "wt_decoder_time_kernel",
    "wt_logmel_create", "wt_logmel_destroy", "wt_logmel_forward", "wt_last_error", "wt_abi_version",
]
DEBUG_EXPORTS = ["wt_dbg_gemm", "wt_dbg_gemm_x3", "wt_dbg_gemm_stamps", "wt_dbg_gemm_f16", "wt_dbg_gemm_f16_variant", "wt_dbg_layernorm", "wt_dbg_encoder_attention", "wt_dbg_encoder_attention_split", "wt_dbg_encoder_attention_x3", "wt_dbg_encoder_attention_f16", "wt_dbg_skinny", "wt_dbg_skinny_gelu_in", "wt_dbg_encoder_attention_occupancy", "wt_dbg_decode_attention",
                 "wt_dbg_decode_attention_folded", "wt_dbg_skinny_pair", "wt_dbg_attention_then_projection", "wt_dbg_self_attention_then_pair",
                 "wt_dbg_skinny_f16", "wt_dbg_decode_attention_f16", "wt_dbg_attention_then_projection_f16", "wt_dbg_skinny_pair_f16", "wt_dbg_gemm_f16_kv"]


class TensorDesc(Structure):
    _fields_ = [("name", c_char * WT_NAME_LEN), ("dtype", c_int32), ("ndim", c_int32), ("shape", c_int64 * WT_MAX_DIMS)]


class Binding(Structure):
    _fields_ = [("name", c_char_p), ("ptr", c_void_p)]


class EngineInfo(Structure):
    _fields_ = [(n, c_int32) for n in ("kind", "precision", "d_model", "n_heads", "n_layers", "ffn_dim", "n_mels",
                                       "max_source_positions", "max_target_positions", "vocab_size")]


class GreedyParams(Structure):
    _fields_ = [("decoder_start_token_id", c_int32), ("eos_token_id", c_int32), ("pad_token_id", c_int32),
                ("max_length", c_int32), ("begin_index", c_int32),
                ("suppress_tokens", POINTER(c_int32)), ("n_suppress_tokens", c_int32),
                ("begin_suppress_tokens", POINTER(c_int32)), ("n_begin_suppress_tokens", c_int32),
                ("forced_decoder_ids", POINTER(c_int32)), ("n_forced", c_int32),
                ("force_eos_step", c_int32), ("logits_trace", c_void_p), ("force_eos_steps", POINTER(c_int32))]


class KernelTimer(Structure):
    _fields_ = [("ms_total", c_float), ("launches", c_int64)]


class EngineLibraryError(RuntimeError):
    pass


_lib = None


def load():
    """Load (once) and return the C-ABI library; raises if it has not been built or cannot be loaded."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EngineLibraryError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            f"(hipcc --offload-arch=gfx950).  whisper-trtllm_amd has no CPU fallback.")
    import torch  # noqa: F401  loads libamdhip64 first so the engine shares torch's HIP runtime (same SONAME)
    try:
        lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    except OSError as exc:
        raise EngineLibraryError(f"cannot load {LIB_PATH}: {exc}") from exc
    for sym in EXPORTS + DEBUG_EXPORTS:
        if not hasattr(lib, sym):
            raise EngineLibraryError(f"{LIB_PATH} does not export {sym}")
    lib.wt_last_error.restype = c_char_p
    lib.wt_abi_version.restype = c_int
    lib.wt_engine_open.argtypes = [c_void_p, c_size_t, c_int, POINTER(c_void_p)]
    lib.wt_engine_clone.argtypes = [c_void_p, POINTER(c_void_p)]
    lib.wt_engine_close.argtypes = [c_void_p]
    lib.wt_engine_close.restype = None
    lib.wt_engine_get_info.argtypes = [c_void_p, POINTER(EngineInfo)]
    lib.wt_engine_gemm_mode.argtypes = [c_void_p]
    lib.wt_engine_infer_shapes.argtypes = [c_void_p, POINTER(TensorDesc), c_int, POINTER(TensorDesc), POINTER(c_int)]
    lib.wt_engine_run.argtypes = [c_void_p, POINTER(Binding), c_int, POINTER(Binding), c_int, c_void_p]
    lib.wt_encoder_forward.argtypes = [c_void_p, c_void_p, c_int, c_void_p, c_void_p]
    lib.wt_decoder_begin.argtypes = [c_void_p, c_void_p, c_int, POINTER(GreedyParams), c_void_p]
    lib.wt_decoder_steps.argtypes = [c_void_p, c_int, c_void_p]
    lib.wt_decoder_poll.argtypes = [c_void_p, POINTER(c_int), POINTER(c_int), POINTER(c_int), c_void_p]
    lib.wt_decoder_run.argtypes = [c_void_p, c_int, POINTER(c_int), POINTER(c_int), c_void_p]
    lib.wt_decoder_read_ids.argtypes = [c_void_p, c_void_p, c_int, c_void_p]
    lib.wt_decoder_greedy.argtypes = [c_void_p, c_void_p, c_int, POINTER(GreedyParams), c_void_p, POINTER(c_int), c_void_p]
    lib.wt_decoder_stream_begin.argtypes = [c_void_p, c_int, c_int, POINTER(GreedyParams), c_void_p]
    lib.wt_decoder_stream_submit.argtypes = [c_void_p, c_void_p, c_int, POINTER(c_int32), POINTER(c_int32), c_void_p]
    lib.wt_decoder_stream_run.argtypes = [c_void_p, c_int, c_int, POINTER(c_int), POINTER(c_int), POINTER(c_int), c_void_p]
    lib.wt_decoder_stream_collect.argtypes = [c_void_p, c_int, POINTER(c_int32), c_int, POINTER(c_int)]
    lib.wt_engine_set_profiling.argtypes = [c_void_p, c_int]
    lib.wt_engine_get_timer.argtypes = [c_void_p, c_char_p, POINTER(KernelTimer)]
    lib.wt_logmel_create.argtypes = [c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int, POINTER(c_void_p)]
    lib.wt_logmel_destroy.argtypes = [c_void_p]
    lib.wt_logmel_destroy.restype = None
    lib.wt_logmel_forward.argtypes = [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]
    lib.wt_decoder_time_cross_attention.argtypes = [c_void_p, c_int, POINTER(c_float), c_void_p]
    lib.wt_decoder_time_kernel.argtypes = [c_void_p, c_char_p, c_int, POINTER(c_float), c_void_p]
    P, I, F = c_void_p, c_int, c_float
    lib.wt_dbg_gemm.argtypes = [P, I, P, P, P, P, I, I, I, I, P]
    lib.wt_dbg_gemm_x3.argtypes = [P, P, P, P, P, I, I, I, I, P, P, I, P]
    lib.wt_dbg_gemm_stamps.argtypes = [P, I, P, P, P, P, I, I, I, I, P, P]
    lib.wt_dbg_gemm_f16.argtypes = [P, I, P, P, P, P, I, I, I, I, I, P]
    lib.wt_dbg_gemm_f16_variant.argtypes = [P, I, P, P, P, P, I, I, I, I, I, I, P]
    lib.wt_dbg_layernorm.argtypes = [P, P, P, P, I, I, P]
    lib.wt_dbg_encoder_attention.argtypes = [P, P, I, I, I, P]
    lib.wt_dbg_encoder_attention_split.argtypes = [P, P, I, I, I, P]
    lib.wt_dbg_encoder_attention_x3.argtypes = [P, P, P, I, I, I, I, P]
    lib.wt_dbg_encoder_attention_f16.argtypes = [P, P, I, I, I, P]
    lib.wt_dbg_skinny.argtypes = [P, P, P, P, P, P, P, I, I, I, I, I, F, P]
    lib.wt_dbg_skinny_gelu_in.argtypes = [P, P, P, P, P, P, P, I, I, I, P]
    lib.wt_dbg_decode_attention.argtypes = [P, P, P, P, P, P, I, I, I, I, I, P]
    lib.wt_dbg_decode_attention_folded.argtypes = [P, P, P, P, P, P, P, P, P, I, I, I, I, I, P]
    lib.wt_dbg_attention_then_projection.argtypes = [P, P, P, P, P, P, P, P, I, I, I, I, I, P]
    lib.wt_dbg_skinny_pair.argtypes = [P, P, P, P, P, I, I, P, P, P, P, P, I, I, I, P]
    lib.wt_dbg_self_attention_then_pair.argtypes = [P, P, P, P, P, P, P, P, P, P, P, I, I, I, I, P]
    lib.wt_dbg_skinny_f16.argtypes = lib.wt_dbg_skinny.argtypes
    lib.wt_dbg_decode_attention_f16.argtypes = [P, P, P, P, P, P, P, P, P, I, I, I, I, I, P]
    lib.wt_dbg_attention_then_projection_f16.argtypes = [P, P, P, P, P, P, P, P, I, I, I, I, P]
    lib.wt_dbg_skinny_pair_f16.argtypes = lib.wt_dbg_skinny_pair.argtypes
    lib.wt_dbg_gemm_f16_kv.argtypes = [P, I, P, P, P, P, I, I, I, I, I, I, P]
    if lib.wt_abi_version() != ABI_VERSION:
        raise EngineLibraryError(f"ABI version mismatch: library {lib.wt_abi_version()}, python {ABI_VERSION}")
    _lib = lib
    return lib


def last_error() -> str:
    return (load().wt_last_error() or b"").decode(errors="replace")


def check(rc: int, what: str):
    if rc != 0:
        raise RuntimeError(f"{what} failed (code {rc}): {last_error()}")
