set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r04d_suite.log 2>&1; echo "rc $?" >> gpurun_out/r04d_suite.log
grep -v amdgpu.ids gpurun_out/r04d_suite.log | tail -6
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r04d_smoke.log 2>&1; echo "smoke rc $?" >> gpurun_out/r04d_smoke.log; tail -3 gpurun_out/r04d_smoke.log
