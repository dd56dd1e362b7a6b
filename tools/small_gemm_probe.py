"""Kernel-duration probe for the tiny decoder GEMMs (run under rocprofv3 --kernel-trace; the launches of
each shape are separated by a marker eltwise launch of a distinctive size so the trace can be cut per shape).
Prints nothing itself about durations: tools/small_gemm_report.py reads the trace."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "future-object-detection_amd"))
import torch
from future_od.native import ops

DEV = "cuda:0"
dt = torch.bfloat16
shapes_nt = [(256, 256, 64), (256, 256, 128), (256, 256, 256), (256, 256, 512), (256, 256, 1024), (256, 256, 2048),
             (256, 2048, 256), (128, 256, 256), (2900, 256, 256)]
shapes_tn = [(32, 256, 256), (64, 256, 256), (128, 256, 256), (256, 256, 256), (512, 256, 256), (256, 2048, 256),
             (256, 256, 2048)]
print("order:", [("nt",) + s for s in shapes_nt] + [("tn",) + s for s in shapes_tn] + [("tn_colsum",) + s for s in shapes_tn[:4]])
for (M, N, K) in shapes_nt:
    a = torch.randn(M, K, device=DEV).to(dt); w = torch.randn(N, K, device=DEV).to(dt)
    b = torch.randn(N, device=DEV)
    for _ in range(12):
        ops.gemm_nt(a, w, shift=b)
    torch.cuda.synchronize()
for cs in (False, True):
    for (M, N1, K2) in (shapes_tn if not cs else shapes_tn[:4]):
        g = torch.randn(M, N1, device=DEV).to(dt); x = torch.randn(M, K2, device=DEV).to(dt)
        dw = torch.zeros(N1, K2, device=DEV); db = torch.zeros(N1, device=DEV)
        for _ in range(12):
            ops.gemm_tn_acc(g, x, dw, colsum=db if cs else None, zeroed=True)
        torch.cuda.synchronize()
