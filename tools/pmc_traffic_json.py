#!/usr/bin/env python3
"""profiles/pmc_traffic_dominant_kernel.json from the two PMC passes of tools/profile_round.sh (FETCH_SIZE, WRITE_SIZE): HBM bytes per
launch of the dominant kernel -- the cross-attention launches of dec_attn_kernel, grid (2, 16, 8) = 256 workgroups at medium.en batch
8 -- with the gfx950 corrections of MI355X_MICROARCH.md §HBM (KiB units, FETCH_SIZE doubled).  The file records the sha256 of
csrc/kernels_decoder.hip it was measured on: bench.py reports `roofline.traffic` only while that still matches the tree."""
import collections
import csv
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def per_launch(path, blocks=256, name="dec_attn_kernel"):
    n, tot, full = 0, 0.0, None
    for r in csv.DictReader(open(path)):
        if name in r["Kernel_Name"] and int(r["Grid_Size"]) // max(1, int(r["Workgroup_Size"])) == blocks:
            n += 1
            tot += float(r["Counter_Value"])
            full = r["Kernel_Name"]
    return n, tot, full


def kernel_source_sha16():
    return hashlib.sha256(open(os.path.join(ROOT, "whisper-trtllm_amd", "csrc", "kernels_decoder.hip"), "rb").read()).hexdigest()[:16]


if __name__ == "__main__":
    fetch_csv, write_csv, out, tag = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4] if len(sys.argv) > 4 else ""
    nf, f, name = per_launch(fetch_csv)
    nw, w, _ = per_launch(write_csv)
    if not nf:
        raise SystemExit("no cross-attention launches (dec_attn_kernel, 256 workgroups) in " + fetch_csv)
    doc = {"kernel": name[:120] + " grid (2,16,8): cross-attention, S=1500, n_split=2, medium.en B=8",
           "fetch_bytes_per_launch": int(round(2 * f * 1024 / nf, -4)), "write_bytes_per_launch": int(round(w * 1024 / max(1, nw), -4)),
           "launches": nf, "algorithmic_bytes_per_launch": 8 * 16 * 1500 * 64 * 4 * 2,
           "method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE in two separate passes over `bench.py --steps 1 --warmup 0 "
                     "--max-length 12 ...` (tools/profile_round.sh pmc); KiB units; FETCH_SIZE doubled (gfx950, MI355X_MICROARCH.md HBM section)",
           "kernels_decoder_sha16": kernel_source_sha16(), "source": tag}
    json.dump(doc, open(out, "w"), indent=1)
    print(json.dumps(doc))
