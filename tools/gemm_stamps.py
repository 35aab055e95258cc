"""Where do the workgroups of one fp32 encoder GEMM launch run, and when?  Probe build of gemm_f32_dma_kernel
(wt_dbg_gemm_stamps): per workgroup its CU (XCC_ID, HW_ID) and wall-clock stamps (100 MHz) at entry, first tile landed,
K loop done, last store issued, stores drained.  Prints workgroups per CU, and the launch timeline in microseconds."""
import os
import sys
from collections import Counter

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import whisper_trtllm_amd as w

lib = w._lib.load()
P = lambda t: t.data_ptr() if t is not None else None
ST = lambda: torch.cuda.current_stream().cuda_stream


def run(M, N, K, resid, warm=3):
    A = torch.randn(M, K, device="cuda")
    W = torch.randn(N, K, device="cuda") * 0.03
    bias = torch.zeros(N, device="cuda")
    C = torch.zeros(M, N, device="cuda")
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    stamps = torch.zeros(tiles, 8, dtype=torch.int64, device="cuda")
    for _ in range(warm):
        lib.wt_dbg_gemm_stamps(P(A), K, P(W), P(bias), P(C) if resid else None, P(C), M, N, K, 0, P(stamps), ST())
    torch.cuda.synchronize()
    s = stamps.cpu().numpy()
    hw, xcc = s[:, 0], s[:, 1] & 0xF
    cu, sh, se = (hw >> 8) & 0xF, (hw >> 12) & 1, (hw >> 13) & 0x7
    where = Counter(zip(xcc.tolist(), se.tolist(), sh.tolist(), cu.tolist()))
    per_cu = Counter(where.values())
    t0 = s[:, 2].min()
    us = (s[:, 2:6] - t0) / 100.0
    print(f"M={M} N={N} K={K} resid={resid}: {tiles} workgroups on {len(where)} CUs; workgroups per CU -> CUs: {dict(sorted(per_cu.items()))}")
    print(f"  per XCD: {dict(sorted(Counter(xcc.tolist()).items()))}")
    q = lambda a: "min %.2f  p10 %.2f  median %.2f  p90 %.2f  max %.2f" % (a.min(), np.percentile(a, 10), np.median(a), np.percentile(a, 90), a.max())
    print("  entry              us:", q(us[:, 0]))
    print("  first tile landed  us:", q(us[:, 1] - us[:, 0]), "(after entry)")
    print("  K loop             us:", q(us[:, 2] - us[:, 1]))
    print("  epilogue, issued   us:", q((s[:, 6] - t0) / 100.0 - us[:, 2]), "(wave 0: last store issued)")
    print("  epilogue + drain   us:", q(us[:, 3] - us[:, 2]))
    print("  exit               us:", q(us[:, 3]))
    loop = us[:, 2] - us[:, 1]
    n_here_ = np.array([where[k] for k in zip(xcc.tolist(), se.tolist(), sh.tolist(), cu.tolist())])
    full = n_here_ == max(per_cu)
    for name, key in (("XCD", xcc), ("SE", se), ("SH", sh), ("CU", cu)):
        print(f"  K loop median us by {name} (CUs holding {max(per_cu)} only):", {int(v): round(float(np.median(loop[full & (key == v)])), 1) for v in sorted(set(key.tolist())) if (full & (key == v)).any()})
    print("  K loop us by workgroup index // 94 (launch order):", [round(float(np.median(loop[i:i + 94])), 1) for i in range(0, tiles, 94)][:16])
    print("  K loop us by tile row by:", "n/a")
    # per-CU load vs finish time
    by_n = {}
    for key, n in where.items():
        m = np.array([(xcc[i], se[i], sh[i], cu[i]) == key for i in range(tiles)]) if False else None
    key_of = list(zip(xcc.tolist(), se.tolist(), sh.tolist(), cu.tolist()))
    for n in sorted(per_cu):
        ends = [us[i, 3] for i in range(tiles) if where[key_of[i]] == n]
        loops = [us[i, 2] - us[i, 1] for i in range(tiles) if where[key_of[i]] == n]
        print(f"  CUs holding {n}: exit median {np.median(ends):.1f} us, K loop median {np.median(loops):.1f} us")


if __name__ == "__main__":
    run(12000, 1024, 1024, True, warm=300)
    run(12000, 4096, 1024, False, warm=100)
    run(12000, 3072, 1024, False, warm=100)
