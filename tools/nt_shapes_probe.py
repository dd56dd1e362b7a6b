"""Kernel durations of the mid-sized NT shapes of the workload (run under rocprofv3 --kernel-trace; 12 launches
per shape, read back with tools/small_gemm_report.py)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "future-object-detection_amd"))
import torch
from future_od.native import ops
DEV = "cuda:0"
dt = torch.bfloat16
shapes = [(14500, 256, 256), (14500, 512, 256), (14500, 2048, 256), (14500, 256, 2048), (2900, 256, 256), (14500, 1536, 256),
          (14500, 256, 3072)]
print("order:", shapes)
for (M, N, K) in shapes:
    a = torch.randn(M, K, device=DEV).to(dt); w = torch.randn(N, K, device=DEV).to(dt)
    b = torch.randn(N, device=DEV)
    for _ in range(12):
        ops.gemm_nt(a, w, shift=b)
    torch.cuda.synchronize()
