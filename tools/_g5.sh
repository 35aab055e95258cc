set -o pipefail
mkdir -p gpurun_out
O=gpurun_out/r4k
export WT_TUNING=1
timeout -k 10 300 python tools/microbench.py gemm_x3 2>&1 | grep -v amdgpu.ids > ${O}_gemm_x3.log || exit 1
cat ${O}_gemm_x3.log
( timeout -k 10 1000 python -m pytest tests -x -q -m gpu > ${O}_suite.log 2>&1; echo "rc $?" >> ${O}_suite.log )
tail -5 ${O}_suite.log
grep -q "rc 0" ${O}_suite.log || exit 1
timeout -k 10 900 python bench.py --steps 3 --warmup 1 > ${O}_bench.json 2> ${O}_bench.err || { tail -20 ${O}_bench.err; exit 1; }
python - <<'PY'
import json
d=json.load(open("gpurun_out/r4k_bench.json"))
c=d["config"]
print("value", d["value"], "ms", d["ms_per_step"], "n32", c["value_n32_decode_steps"], "varlen", c["value_varlen"], "cont", c["value_varlen_continuous"], "b16", c["value_batch16_per_gpu"])
print("enc", d["roofline_encoder"])
print("long", d["varlen"]["long_run"])
PY
