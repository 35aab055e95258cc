#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes per kernel (+grid) as HBM bytes per launch.
gfx950 corrections (MI355X_MICROARCH.md §HBM): the counters are in KiB; FETCH_SIZE reports exactly half of the
bytes of a wide coalesced streaming read, so it is doubled; WRITE_SIZE is exact for 16-B-per-lane stores."""
import collections
import csv
import sys


def load(path):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        key = (r["Kernel_Name"][:60], int(r["Grid_Size"]) // max(1, int(r["Workgroup_Size"])))
        agg[key][0] += 1
        agg[key][1] += float(r["Counter_Value"])
    return agg


fetch, write = load(sys.argv[1]), load(sys.argv[2])
print(f"{'kernel':<62}{'blocks':>8}{'n':>6}{'fetch MB (x2 corrected)':>26}{'write MB':>12}")
for k in sorted(fetch, key=lambda k: -fetch[k][1]):
    n, f = fetch[k]
    w = write.get(k, [1, 0.0])
    print(f"{k[0]:<62}{k[1]:>8}{n:>6}{2 * f * 1024 / n / 1e6:>26.2f}{w[1] * 1024 / max(1, w[0]) / 1e6:>12.2f}")
