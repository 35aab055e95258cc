#!/usr/bin/env python3
"""MFMA utilisation and effective clock per kernel from ONE `rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES
GRBM_GUI_ACTIVE` pass: usage pmc_mfma.py <counter_collection.csv> <kernel_trace.csv>.
GRBM_GUI_ACTIVE is summed over the 8 XCDs (MI355X_MICROARCH.md, DVFS give-back): clock = GUI/8/duration;
SQ_VALU_MFMA_BUSY_CYCLES is per SIMD, summed over the 1024 SIMDs: utilisation = busy / (1024 * GUI/8)."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
trace = {r["Dispatch_Id"]: r for r in csv.DictReader(open(sys.argv[2]))}
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for r in rows:
    k = (r["Kernel_Name"][:48], int(r["Grid_Size"]) // int(r["Workgroup_Size"]))
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    t = trace.get(r["Dispatch_Id"])
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE" and t:
        agg[k]["ns"] += int(t["End_Timestamp"]) - int(t["Start_Timestamp"])
        cnt[k] += 1
print(f"{'kernel':<50}{'blocks':>7}{'n':>6}{'avg us':>10}{'clock GHz':>11}{'MFMA busy':>11}{'x157.3 TF':>11}")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1]["ns"])[:10]:
    ns, gui, mf = v["ns"], v["GRBM_GUI_ACTIVE"], v["SQ_VALU_MFMA_BUSY_CYCLES"]
    if not ns or not gui:
        continue
    util = mf / (1024 * (gui / 8))
    print(f"{k[0]:<50}{k[1]:>7}{cnt[k]:>6}{ns / cnt[k] / 1e3:>10.1f}{gui / 8 / ns:>11.2f}{util:>11.3f}{util * 157.3:>11.1f}")
