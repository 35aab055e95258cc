set -o pipefail
mkdir -p gpurun_out
( timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r4g_suite.log 2>&1; echo "rc $?" >> gpurun_out/r4g_suite.log )
tail -6 gpurun_out/r4g_suite.log
grep -q "rc 0" gpurun_out/r4g_suite.log || exit 1
bash tools/profile_round.sh r04a bench
