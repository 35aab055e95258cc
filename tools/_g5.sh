set -o pipefail
mkdir -p gpurun_out
O=gpurun_out/r4n
timeout -k 10 900 python bench.py --steps 3 --warmup 1 > ${O}_bench.json 2> ${O}_bench.err || { tail -20 ${O}_bench.err; exit 1; }
python - <<'PY'
import json
d=json.load(open("gpurun_out/r4n_bench.json"))
c=d["config"]
print("value", d["value"], "ms", d["ms_per_step"], "n32", c["value_n32_decode_steps"], "varlen", c["value_varlen"], "cont", c["value_varlen_continuous"], "b16", c["value_batch16_per_gpu"], "fp16", c["value_fp16_decoder_b16"])
print("enc", d["roofline_encoder"])
print("decode", d["roofline_decode"]["ms_per_step"], "cross", d["roofline"]["avg_launch_us"])
print("long", d["varlen"]["long_run"], d["varlen"].get("continuous_workers"), d["varlen"].get("length_sorted_workers"))
PY
WT_TUNING=1 WT_ATTN_X3=0 timeout -k 10 400 python bench.py --steps 3 --warmup 1 --no-varlen --no-batch16 --no-cpu-baseline > ${O}_bench_attn_native.json 2>/dev/null
python - <<'PY'
import json
d=json.load(open("gpurun_out/r4n_bench_attn_native.json"))
print("attention on the fp32 MFMA: value", d["value"], "n32", d["config"]["value_n32_decode_steps"], d["roofline_encoder"]["enc_attn_total_ms"], d["roofline_encoder"]["total_ms"])
PY
