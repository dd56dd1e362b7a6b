"""Is the start of the backward pass host-bound?  A GPU-side delay of D ms is inserted right after the matcher's
host sync (before the loss kernels).  If the step grows by D the GPU was never waiting for the host there; if it grows
by less, the difference is time the GPU used to sit idle while the host issued the decoder / encoder backward."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "future-object-detection_amd"))
import torch
from types import SimpleNamespace
import bench
from future_od.datasets.synthetic import make_batch
from future_od.optim import FusedAdamW
from future_od.models import set_criterion as SC

dev = torch.device("cuda", 0)
model, detr = bench.build(SimpleNamespace(), dev, False, 5, "bf16")
model.eval()
opt = FusedAdamW(model.parameters(), lr=1e-4, weight_decay=1e-4, max_norm=0.1)
data = make_batch(2, 6, 900, 1600, seed=1234, device=dev)
# calibrate torch.cuda._sleep
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda._sleep(1000000); torch.cuda.synchronize()
e0.record(); torch.cuda._sleep(10000000); e1.record(); torch.cuda.synchronize()
cyc_per_ms = 10000000 / e0.elapsed_time(e1)
DELAY = [0.0]
orig = SC._SetLossFn.forward


def patched(ctx, *a):
    if DELAY[0] > 0:
        torch.cuda._sleep(int(DELAY[0] * cyc_per_ms))
    return orig(ctx, *a)


SC._SetLossFn.forward = staticmethod(patched)
# other insertion points: the start of the step, the start of the backbone's backward (POINT=step|backbone)
POINT = os.environ.get("POINT", "matcher")
if POINT != "matcher":
    SC._SetLossFn.forward = staticmethod(orig)
if POINT == "backbone":
    from future_od.native import backbone as BB
    orig_bb = BB.BackboneFn.backward

    def patched_bb(ctx, *a):
        if DELAY[0] > 0:
            torch.cuda._sleep(int(DELAY[0] * cyc_per_ms))
        return orig_bb(ctx, *a)

    BB.BackboneFn.backward = staticmethod(patched_bb)


def step():
    if POINT == "step" and DELAY[0] > 0:
        torch.cuda._sleep(int(DELAY[0] * cyc_per_ms))
    opt.zero_grad()
    out, _s, loss, stats, od = model(data=data, distributed=False)
    loss.backward()
    opt.step()


def run(delay, n=6):
    DELAY[0] = delay
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


for _ in range(2):
    step()
base = run(0.0)
print(f"no delay: {base:.2f} ms/step")
for d in (2.0, 4.0, 8.0):
    t = run(d)
    print(f"delay {d:.0f} ms at {POINT}: {t:.2f} ms/step (+{t - base:.2f}): {d - (t - base):.2f} ms absorbed")
