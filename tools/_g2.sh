set -o pipefail
mkdir -p gpurun_out
O=gpurun_out/r4c
( timeout -k 10 900 python -m pytest tests/test_gpu_stream.py tests/test_gpu_session.py -x -q -k "stream or continuous or cal_wer or ragged or natural or tiny_en or fp16_decoder_engine or small_pool or run_py" > ${O}_tests.log 2>&1; echo "rc $?" >> ${O}_tests.log )
tail -15 ${O}_tests.log
grep -q "rc 0" ${O}_tests.log || exit 1
timeout -k 10 900 python bench.py --steps 3 --warmup 1 > ${O}_bench.json 2> ${O}_bench.err || { tail -20 ${O}_bench.err; exit 1; }
python - <<'PY'
import json
d=json.load(open("gpurun_out/r4c_bench.json"))
print("value", d["value"], "n32", d["config"]["value_n32_decode_steps"], "varlen", json.dumps(d["varlen"], indent=0)[:1500])
print("decode", d["roofline_decode"]["ms_per_step"], "enc", d["roofline_encoder"]["frac"], d["roofline_encoder"]["enc_attn_tflops"], "traffic", d["roofline"]["traffic"], d["roofline"]["traffic_from"])
PY
