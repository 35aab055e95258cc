#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV: per (kernel, grid) count / avg / total."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    key = (r["Kernel_Name"][:60], int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), r["Grid_Size_Y"], r["Grid_Size_Z"])
    agg[key][0] += 1
    agg[key][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
tot = sum(v[1] for v in agg.values())
print(f"total kernel time {tot/1e3:.2f} ms over {len(rows)} dispatches")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"{k[0]:<62} blocks=({k[1]},{k[2]},{k[3]}) n={v[0]:<6} avg={v[1]/v[0]:8.2f} us total={v[1]/1e3:9.2f} ms {100*v[1]/tot:5.1f}%")
