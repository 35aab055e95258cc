set -o pipefail
mkdir -p gpurun_out
{
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "attention" 2>&1 | tail -3
timeout -k 10 300 python tools/microbench.py enc_attn 2>&1 | grep -i "enc_attn"
} > gpurun_out/r04c_attn_tr.log 2>&1
cat gpurun_out/r04c_attn_tr.log
