#!/usr/bin/env python3
"""VGPRs / AGPRs / scratch / occupancy / LDS of every kernel of one csrc file (hipcc cross-compiles without a GPU):
    python tools/kernel_resources.py kernels_decoder.hip [name-filter]"""
import os, re, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if os.path.exists(sys.argv[1]) else os.path.join(ROOT, "whisper-trtllm_amd", "csrc", sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
r = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-c", src, "-o", "/tmp/_kres.o",
                    "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
if r.returncode:
    sys.exit(r.stderr[-3000:])
cxxfilt = shutil.which("c++filt") or shutil.which("llvm-cxxfilt")
cur, rows = {}, []
for line in r.stderr.splitlines():
    for key, pat in (("name", r"Function Name: (\S+)"), ("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                     ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
        m = re.search(pat, line)
        if m:
            cur[key] = m.group(1)
            if key == "lds":
                rows.append(cur); cur = {}
names = [x["name"] for x in rows]
if cxxfilt:
    names = subprocess.run([cxxfilt], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
for x, n in zip(rows, names):
    n = re.sub(r"^void (wt::)?", "", n).split("(")[0]
    if flt in n:
        print(f"vgpr={x.get('vgpr'):>3} agpr={x.get('agpr'):>3} scratch={x.get('scratch'):>3} occ={x.get('occ')} lds={x.get('lds'):>6}  {n}")
