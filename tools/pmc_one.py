"""Run one op repeatedly (for rocprofv3 --pmc): python tools/pmc_one.py tn|tnconv|nt"""
import os, sys, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "future-object-detection_amd"))
import torch
from future_od.native import ops
DEV = "cuda:0"; dt = torch.bfloat16
which = sys.argv[1]
if which == "tn":
    M, N1, K2 = 14500, 256, 256
    g = torch.randn(M, N1, device=DEV).to(dt); x = torch.randn(M, K2, device=DEV).to(dt); dw = torch.zeros(N1, K2, device=DEV)
    f = lambda: ops.gemm_tn_acc(g, x, dw)
elif which == "tnconv":
    n, h, w, cin, cout, k, s, p = 10, 57, 100, 256, 256, 3, 1, 1
    x = torch.randn(n, h, w, cin, device=DEV).to(dt); geom = ops.conv_geom(x.shape, cout, k, s, p)
    dy = torch.randn(n, geom.Ho, geom.Wo, cout, device=DEV).to(dt); dw = torch.zeros(cout, k, k, cin, device=DEV)
    f = lambda: ops.conv2d_wgrad_acc(dy, x, dw, geom)
else:
    n, h, w, cin, cout, k, s, p = 10, 57, 100, 256, 256, 3, 1, 1
    x = torch.randn(n, h, w, cin, device=DEV).to(dt); geom = ops.conv_geom(x.shape, cout, k, s, p)
    wt = torch.randn(cout, k, k, cin, device=DEV).to(dt); sh = torch.zeros(cout, device=DEV)
    f = lambda: ops.conv2d_fwd(x, wt, geom, shift=sh, relu=True)
for _ in range(5):
    f()
torch.cuda.synchronize()
