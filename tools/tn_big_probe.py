"""Weight-gradient kernel probe: a few backbone layers and encoder shapes with the 8-wave LDS-DMA kernel under
forced split counts and both block orders, against the 128 x 128 kernel (time in us, TFLOP/s)."""
import os, sys, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "future-object-detection_amd"))
import torch
from future_od.native import ops

DEV = "cuda:0"
dt = torch.bfloat16


def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / iters


def setenv(**kw):
    for k, v in kw.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = str(v)


LAYERS = [("layer2.0.conv1", 10, 225, 400, 256, 128, 1, 1, 0), ("layer4.0.conv3", 10, 29, 50, 512, 2048, 1, 1, 0),
          ("layer4.1.conv1", 10, 29, 50, 2048, 512, 1, 1, 0), ("layer2.1.conv1", 10, 113, 200, 512, 128, 1, 1, 0),
          ("layer3.1.conv2", 10, 57, 100, 256, 256, 3, 1, 1), ("layer2.1.conv2", 10, 113, 200, 128, 128, 3, 1, 1),
          ("layer4.1.conv2", 10, 29, 50, 512, 512, 3, 1, 1), ("layer3.1.conv1", 10, 57, 100, 1024, 256, 1, 1, 0),
          ("layer2.0.conv3", 10, 113, 200, 128, 512, 1, 1, 0), ("layer3.0.conv3", 10, 57, 100, 256, 1024, 1, 1, 0)]
for name, n, h, w, cin, cout, k, s, p in LAYERS:
    x = torch.randn(n, h, w, cin, device=DEV).to(dt)
    g = ops.conv_geom(x.shape, cout, k, s, p)
    dy = torch.randn(n, g.Ho, g.Wo, cout, device=DEV).to(dt)
    dw = torch.zeros(cout, k, k, cin, device=DEV)
    fl = 2.0 * n * g.Ho * g.Wo * cout * k * k * cin
    run = lambda: ops.conv2d_wgrad_acc(dy, x, dw, g)
    setenv(FOD_TN_BIG=0, FOD_TN_BIG_SPLITS=None, FOD_TN_XCD=None)
    t = timeit(run)
    out = [f"{name:16s} 128x128 {t * 1e6:6.1f} us {fl / t / 1e12:6.1f} TF |"]
    setenv(FOD_TN_BIG=2, FOD_TN_WS=0)
    t = timeit(run)
    out.append(f"big, atomics {t * 1e6:6.1f} us {fl / t / 1e12:6.1f} TF |")
    setenv(FOD_TN_BIG=2, FOD_TN_WS=None)
    t = timeit(run)
    out.append(f"big, partial tiles + reduce {t * 1e6:6.1f} us {fl / t / 1e12:6.1f} TF")
    setenv(FOD_TN_BIG=None)
    print(" ".join(out), flush=True)
for M, N1, K2 in [(14500, 2048, 256), (14500, 256, 2048), (14500, 768, 256), (14500, 256, 256)]:
    gg = torch.randn(M, N1, device=DEV).to(dt)
    xx = torch.randn(M, K2, device=DEV).to(dt)
    dw = torch.zeros(N1, K2, device=DEV)
    fl = 2.0 * M * N1 * K2
    run = lambda: ops.gemm_tn_acc(gg, xx, dw)
    setenv(FOD_TN_BIG=0)
    t0 = timeit(run)
    setenv(FOD_TN_BIG=2, FOD_TN_WS=0)
    t1 = timeit(run)
    setenv(FOD_TN_BIG=2, FOD_TN_WS=None)
    t2 = timeit(run)
    setenv(FOD_TN_BIG=None)
    print(f"dense {M}x{N1}x{K2}: 128x128 {t0 * 1e6:6.1f} us {fl / t0 / 1e12:6.1f} TF | big, atomics {t1 * 1e6:6.1f} us | big, partial tiles + reduce {t2 * 1e6:6.1f} us {fl / t2 / 1e12:6.1f} TF", flush=True)
