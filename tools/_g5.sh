set -o pipefail
mkdir -p gpurun_out
O=gpurun_out/r4o
( timeout -k 10 1000 python -m pytest tests -x -q -m gpu > ${O}_suite.log 2>&1; echo "rc $?" >> ${O}_suite.log )
tail -5 ${O}_suite.log
grep -q "rc 0" ${O}_suite.log || exit 1
timeout -k 10 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > ${O}_bench.json 2> ${O}_bench.err || { tail -20 ${O}_bench.err; exit 1; }
python - <<'PY'
import json
d=json.load(open("gpurun_out/r4o_bench.json"))
c=d["config"]
print("value", d["value"], "ms", d["ms_per_step"], "n32", c["value_n32_decode_steps"], "cont", c["value_varlen_continuous"])
print("enc", d["roofline_encoder"]["total_ms"], d["roofline_encoder"]["enc_attn_total_ms"], d["roofline_encoder"]["useful_fp32_tflops"])
PY
