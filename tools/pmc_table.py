#!/usr/bin/env python3
"""Per (kernel, workgroups) mean of every counter in a rocprofv3 --pmc counter_collection.csv (any counter set)."""
import collections
import csv
import sys

agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.Counter())
for r in csv.DictReader(open(sys.argv[1])):
    k = (r["Kernel_Name"][:46], int(r["Grid_Size"]) // max(1, int(r["Workgroup_Size"])))
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[k][r["Counter_Name"]] += 1
names = sorted({c for v in agg.values() for c in v})
print(f"{'kernel':<48}{'wgs':>7}{'n':>5}" + "".join(f"{c[-18:]:>20}" for c in names))
for k in sorted(agg, key=lambda k: -sum(agg[k].values()))[: int(sys.argv[2]) if len(sys.argv) > 2 else 14]:
    n = max(cnt[k].values())
    print(f"{k[0]:<48}{k[1]:>7}{n:>5}" + "".join(f"{agg[k][c] / max(1, cnt[k][c]):>20.4g}" for c in names))
