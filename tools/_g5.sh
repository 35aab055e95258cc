set -o pipefail
mkdir -p gpurun_out
O=gpurun_out/r4p
export WT_TUNING=1
( timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "gemm_x3 or attention_x3" > ${O}_t.log 2>&1; echo "rc $?" >> ${O}_t.log ); tail -3 ${O}_t.log
grep -q "rc 0" ${O}_t.log || { tail -30 ${O}_t.log; exit 1; }
for prio in 0 1; do echo "== WT_GEMM_X3_PRIO=$prio" >> ${O}_gemm.log; WT_GEMM_X3_PRIO=$prio timeout -k 10 300 python tools/microbench.py gemm_x3 2>&1 | grep -v amdgpu.ids >> ${O}_gemm.log || exit 1; done
cat ${O}_gemm.log
timeout -k 10 200 python tools/microbench.py enc_attn 2>&1 | grep "enc_attn" | tee ${O}_attn.log
