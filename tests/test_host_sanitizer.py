"""CPU: the C-ABI's host-only logic (blob parser, config limits, Session-style shape inference -- csrc/host_logic.cpp) built with
g++ -fsanitize=address,undefined and driven with targeted corruptions + random mutations of valid engine blobs
(tests/native/fuzz_host_logic.cpp).  The GPU never enters: no HIP in that translation unit."""
import os
import shutil
import subprocess

import pytest

import whisper_trtllm_amd as w

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def fuzzer(tmp_path_factory):
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("g++ not available")
    out = tmp_path_factory.mktemp("san") / "fuzz_host_logic"
    cmd = [gxx, "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer",
           os.path.join(ROOT, "tests", "native", "fuzz_host_logic.cpp"),
           os.path.join(ROOT, "whisper-trtllm_amd", "csrc", "host_logic.cpp"), "-o", str(out)]
    subprocess.run(cmd, check=True)
    return str(out)


@pytest.mark.parametrize("kind,precision", [("encoder", "float32"), ("encoder", "float16"), ("decoder", "float32"), ("decoder", "float16")])
def test_blob_parser_and_shape_inference_under_asan_ubsan(fuzzer, tmp_path, kind, precision):
    cfg = w.synthetic.get_config("toy-short")
    weights = w.synthetic.make_weights(cfg, 3)
    blob = (w.convert.build_encoder_engine(cfg, weights, precision=precision) if kind == "encoder"
            else w.convert.build_decoder_engine(cfg, weights, precision=precision))
    path = tmp_path / f"{kind}.engine"
    path.write_bytes(blob)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([fuzzer, str(path), "3000", "12345"], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "0 failures" in r.stdout and "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stdout + r.stderr
