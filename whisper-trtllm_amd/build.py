"""Build the HIP C-ABI library in-tree: whisper-trtllm_amd/lib/libwhisper_trtllm_amd.so (gfx950 only).

hipcc cross-compiles without a GPU, so this runs in the build container; the built .so travels to the
GPU box with the repo snapshot (it is git-ignored, not gpurun-ignored).
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libwhisper_trtllm_amd.so")
SOURCES = ["engine.hip", "host_logic.cpp", "frontend.hip", "debug_api.hip", "kernels_encoder.hip", "kernels_encoder_f16.hip", "kernels_decoder.hip"]
HEADERS = ["wt_common.h", "host_logic.h", os.path.join("..", "..", "include", "whisper_trtllm_amd.h"),
           os.path.join("..", "..", "include", "whisper_trtllm_amd_debug.h")]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: cannot build libwhisper_trtllm_amd.so")


def is_stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS]
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force: bool = False, verbose: bool = False) -> str:
    if not force and not is_stale():
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    objs, cmds = [], []
    for src in SOURCES:
        obj = os.path.join(LIB_DIR, os.path.splitext(src)[0] + ".o")
        cmds.append([_hipcc(), "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-Wall", "-Wno-unused-result", "-Wno-unused-value",
                     "-c", os.path.join(CSRC, src), "-o", obj])
        objs.append(obj)
    # translation units are independent: compile them side by side (the three kernel files take ~25 s each)
    procs = []
    for cmd in cmds:
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        procs.append(subprocess.Popen(cmd))
    failed = [cmd for cmd, pr in zip(cmds, procs) if pr.wait() != 0]
    if failed:
        raise subprocess.CalledProcessError(1, failed[0])
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + objs
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    return LIB_PATH


def build_c_example(verbose: bool = False) -> str:
    """examples/c/wt_greedy: a plain C/C++ host of the C-ABI (no Python, no torch), linked against the library above."""
    root = os.path.dirname(HERE)
    src, exe = os.path.join(root, "examples", "c", "wt_greedy.cpp"), os.path.join(root, "examples", "c", "wt_greedy")
    newest = max(os.path.getmtime(src), os.path.getmtime(os.path.join(root, "include", "whisper_trtllm_amd.h")), os.path.getmtime(LIB_PATH))
    if os.path.exists(exe) and os.path.getmtime(exe) >= newest:
        return exe
    cmd = [_hipcc(), "-O2", "-pthread", "-I" + os.path.join(root, "include"), src, "-L" + LIB_DIR, "-lwhisper_trtllm_amd",
           "-Wl,-rpath,$ORIGIN/../../whisper-trtllm_amd/lib", "-o", exe]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    return exe


def build_mfma_probe(verbose: bool = False) -> str:
    """tools/probes/mfma_power/mfma_power: a bare bf16 MFMA loop (operands in registers, no memory traffic).  `bench.py` runs it with
    --quick for the rate the matrix pipe of THIS device sustains on random operands under its power management -- the practical ceiling
    the x3 GEMM's TFLOP/s are shown beside."""
    d = os.path.join(os.path.dirname(HERE), "tools", "probes", "mfma_power")
    src, exe = os.path.join(d, "mfma_power.hip"), os.path.join(d, "mfma_power")
    if os.path.exists(exe) and os.path.getmtime(exe) >= os.path.getmtime(src):
        return exe
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-w", "-o", exe, src]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    return exe


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
    print(build_c_example(verbose=True))
    print(build_mfma_probe(verbose=True))
