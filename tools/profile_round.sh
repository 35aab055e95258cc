#!/bin/bash
# One gpurun call = the whole profiling recipe of a round (profiles/README.md).  Usage (from the repo root, on the GPU box):
#   bash tools/profile_round.sh r03b [bench|trace|pmc|all]
# Writes gpurun_out/<tag>_*; copy the summaries you want judged into profiles/.  The three phases fit one gpurun call each (since
# round 3 a bench line also runs the variable-length and fp16-engine legs; traces and PMC passes skip those two).
set -o pipefail
TAG=${1:-rXX}
PHASE=${2:-all}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
step() { echo "[profile_round] $*"; }

if [ "$PHASE" = all ] || [ "$PHASE" = bench ]; then
# ---- plain bench lines (no profiler): headline config, then BASELINE configs 2, 3, 4
step "bench medium.en B=8 (headline)"
timeout -k 10 400 python3 $R/bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err || exit 1
step "bench config 4 (medium.en fp16 encoder + fp32 decoder, B=16)"
timeout -k 10 400 python3 $R/bench.py --encoder-precision float16 --batch 16 --no-batch16 --no-cpu-baseline > $O/${TAG}_cfg4_bench.json 2> $O/${TAG}_cfg4_bench.err || exit 1
step "bench config 3 (small.en B=8)"
timeout -k 10 300 python3 $R/bench.py --model whisper-small.en --no-batch16 --no-cpu-baseline > $O/${TAG}_cfg3_bench.json 2> $O/${TAG}_cfg3_bench.err || exit 1
step "bench config 2 (tiny.en B=1)"
timeout -k 10 300 python3 $R/bench.py --model whisper-tiny.en --batch 1 --no-batch16 --no-cpu-baseline > $O/${TAG}_cfg2_bench.json 2> $O/${TAG}_cfg2_bench.err || exit 1

fi
# ---- kernel traces (rocprofv3 --kernel-trace --stats), summaries by tools/prof_summary.py
trace() {  # name, bench args...
    local name=$1; shift
    export WT_SAVE_MAPS=$O/${TAG}_${name}_proc_maps.txt   # load addresses of every module of the profiled process: any crash trace of this run stays symbolisable
    step "kernel trace $name"
    rm -rf $O/${TAG}_trace_$name
    timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_trace_$name -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-batch16 --no-varlen --no-fp16-decoder "$@" > $O/${TAG}_${name}_bench_under_rocprof.json 2> $O/${TAG}_trace_$name.err || return 1
    local kt=$(find $O/${TAG}_trace_$name -name "*kernel_trace.csv" | head -1)
    local ks=$(find $O/${TAG}_trace_$name -name "*kernel_stats.csv" | head -1)
    python3 $R/tools/prof_summary.py $kt > $O/${TAG}_${name}_kernel_trace_summary.txt
    cp $ks $O/${TAG}_${name}_kernel_stats.csv
    rm -rf $O/${TAG}_trace_$name      # the raw trace is tens of MB: keep the summaries only
}
if [ "$PHASE" = all ] || [ "$PHASE" = trace ]; then
trace medium || exit 1
trace cfg4 --encoder-precision float16 --batch 16 || exit 1
trace cfg3 --model whisper-small.en || exit 1
trace cfg2 --model whisper-tiny.en --batch 1 || exit 1
fi

# ---- PMC passes (each counter set in its own run, kernel-trace only: the pool refuses --pmc with other trace domains)
pmc() {  # name, counters..., --, bench args
    local name=$1; shift
    local ctr=()
    while [ "$1" != "--" ]; do ctr+=("$1"); shift; done
    shift
    step "pmc $name: ${ctr[*]}"
    rm -rf $O/${TAG}_pmc_$name
    timeout -k 10 600 rocprofv3 --kernel-trace --pmc "${ctr[@]}" --output-format csv -d $O/${TAG}_pmc_$name -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-batch16 --no-roofline --no-varlen --no-fp16-decoder "$@" > /dev/null 2> $O/${TAG}_pmc_$name.err || return 1
}
if [ "$PHASE" = all ] || [ "$PHASE" = pmc ]; then
pmc fetch FETCH_SIZE -- --max-length 12 || exit 1
pmc write WRITE_SIZE -- --max-length 12 || exit 1
python3 $R/tools/pmc_summary.py $(find $O/${TAG}_pmc_fetch -name "*counter_collection.csv" | head -1) $(find $O/${TAG}_pmc_write -name "*counter_collection.csv" | head -1) > $O/${TAG}_pmc_fetch_write_per_launch.txt
python3 $R/tools/pmc_traffic_json.py $(find $O/${TAG}_pmc_fetch -name "*counter_collection.csv" | head -1) $(find $O/${TAG}_pmc_write -name "*counter_collection.csv" | head -1) $O/${TAG}_pmc_traffic_dominant_kernel.json "profiles/${TAG}_pmc_fetch_write_per_launch.txt" || exit 1
pmc mfma SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -- --max-length 4 || exit 1
python3 $R/tools/pmc_mfma.py $(find $O/${TAG}_pmc_mfma -name "*counter_collection.csv" | head -1) $(find $O/${TAG}_pmc_mfma -name "*kernel_trace.csv" | head -1) > $O/${TAG}_pmc_mfma_busy.txt
pmc mfma16 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -- --max-length 4 --encoder-precision float16 --batch 16 || exit 1
python3 $R/tools/pmc_mfma.py $(find $O/${TAG}_pmc_mfma16 -name "*counter_collection.csv" | head -1) $(find $O/${TAG}_pmc_mfma16 -name "*kernel_trace.csv" | head -1) > $O/${TAG}_cfg4_pmc_mfma_busy.txt
rm -rf $O/${TAG}_pmc_fetch $O/${TAG}_pmc_write $O/${TAG}_pmc_mfma $O/${TAG}_pmc_mfma16
fi
step done
ls -la $O | grep ${TAG}_
