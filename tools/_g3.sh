set -o pipefail
mkdir -p gpurun_out
O=$GRAFT_REPO_ROOT/gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
export WT_TUNING=1 WT_ENC_ATTN_VAR=0
rocprofv3 --list-avail 2>/dev/null | grep -oE "SQ_[A-Z_0-9]+|LDS[A-Za-z_]*" | sort -u | tr '\n' ' ' > $O/r4f_counters.txt
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" "SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_VALU" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rm -rf /tmp/pmc_$tag
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pmc_$tag -- python3 $R/tools/microbench.py enc_attn > /dev/null 2> /tmp/pmc_$tag.err || { tail -5 /tmp/pmc_$tag.err; continue; }
  f=$(find /tmp/pmc_$tag -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY' >> $O/r4f_pmc.txt
import csv, sys, collections
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    if "enc_attn_kernel" in r["Kernel_Name"] and "f16" not in r["Kernel_Name"]:
        k = (r["Kernel_Name"][:40], r["Counter_Name"])
        agg[k][0] += 1; agg[k][1] += float(r["Counter_Value"])
for k, v in sorted(agg.items()):
    print(f"{k[0]:42s}{k[1]:34s} n={v[0]:3d} avg={v[1]/v[0]:.4g}")
PY
done
cat $O/r4f_pmc.txt
