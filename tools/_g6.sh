set -o pipefail
bash tools/profile_round.sh r04b all || exit 1
cd /tmp && export TMPDIR=/tmp
# one multi-worker run under the profiler must exit cleanly (WhisperPipeline clamps itself to one worker)
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/r04b_two_workers -- python3 $GRAFT_REPO_ROOT/tools/two_workers.py whisper-tiny.en 4 8 1,4 > $GRAFT_REPO_ROOT/gpurun_out/r04b_workers_under_rocprof.txt 2>&1; echo "exit code $?" >> $GRAFT_REPO_ROOT/gpurun_out/r04b_workers_under_rocprof.txt
tail -5 $GRAFT_REPO_ROOT/gpurun_out/r04b_workers_under_rocprof.txt
