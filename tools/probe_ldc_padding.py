"""Is the tall GEMM's output write bound by the 4 KB row stride of a [14500, 2048] bf16 tensor?  The same launch with the
output's leading dimension padded (2048 -> 2048 + 64 columns), stream kernel and tiled kernel."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "future-object-detection_amd"))
import torch
from future_od.native import lib as L, ops


def t(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(it): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / it * 1e6


M, K = 14500, 256
for N in (2048, 1024, 512):
    a = torch.randn(M, K, device="cuda").bfloat16(); b = torch.randn(N, K, device="cuda").bfloat16()
    s = torch.randn(N, device="cuda")
    for pad in (0, 64, 32, 8):
        ldc = N + pad
        out = torch.empty(M, ldc, device="cuda", dtype=torch.bfloat16)
        e = ops._epi(None, s, None, 0, 0, None, 0, True)
        res = []
        for v in ("1", "0"):
            os.environ["FOD_NT_STREAM"] = v
            res.append(min(t(lambda: L.call("fod_gemm_nt", L.BF16, a.data_ptr(), K, 0, b.data_ptr(), K, out.data_ptr(), ldc, M, N, K, e, ops.stream())) for _ in range(3)))
        print(f"N {N:5d} ldc {ldc:5d}: stream {res[0]:6.1f} us   tiled {res[1]:6.1f} us")
