import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import whisper_trtllm_amd as w
lib = w._lib.load()
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
ST = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
B = 8
for (N, K, xmode) in ((4096, 1024, 1 | 4), (1024, 4096, 0 | 4), (1024, 1024, 4), (3072, 1024, 1 | 4)):
    L = 24
    W = torch.randn(L, N, K, device="cuda") * 0.02
    X = torch.randn(B, K, device="cuda"); g = torch.ones(K, device="cuda"); be = torch.zeros(K, device="cuda"); bias = torch.zeros(N, device="cuda")
    Y = torch.empty(B, N, device="cuda")
    gr = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for i in range(3): lib.wt_dbg_skinny(P(X), P(g), P(be), P(W[i]), P(bias), None, P(Y), B, N, K, xmode, 0, 1.0, ST())
        torch.cuda.synchronize()
        with torch.cuda.graph(gr, stream=s):
            for i in range(L): lib.wt_dbg_skinny(P(X), P(g), P(be), P(W[i]), P(bias), None, P(Y), B, N, K, xmode, 0, 1.0, ST())
    for _ in range(3): gr.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); [gr.replay() for _ in range(10)]; e1.record(); torch.cuda.synchronize()
    per = e0.elapsed_time(e1) / 10 / L * 1e3
    buf = np.zeros(8192 * 4, dtype=np.uint64)
    lib.wt_dbg_read_stamps.argtypes = [ctypes.c_void_p]
    assert lib.wt_dbg_read_stamps(buf.ctypes.data_as(ctypes.c_void_p)) == 0
    st = buf.reshape(-1, 4).astype(np.int64)
    nw = {4096: 2048, 3072: 1536}.get(N, 512 if K == 1024 else 2048)
    st = st[:nw]
    base = st[:, 0].min()
    r = (st - base) * 0.01   # us (100 MHz)
    q = lambda a: f"min {a.min():5.2f} p50 {np.median(a):5.2f} p90 {np.percentile(a, 90):5.2f} max {a.max():5.2f}"
    print(f"N={N} K={K}: {per:.2f} us/launch; waves {nw}")
    print("  start   ", q(r[:, 0])); print("  X ready ", q(r[:, 1])); print("  W ready ", q(r[:, 2])); print("  end     ", q(r[:, 3]))
    print("  per-wave W wait (tW - t0)", q(r[:, 2] - r[:, 0]), " tail (tE - tW)", q(r[:, 3] - r[:, 2]))
