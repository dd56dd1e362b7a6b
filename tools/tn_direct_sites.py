"""Which weight gradients of a headline step still launch fod_gemm_tn_acc one by one (not through the queues): shapes,
counts and call sites.  Runs the step eagerly with the queue enabled (FOD_WGRAD_QUEUE_EAGER=1), as a capture would."""
import collections, os, sys, traceback
os.environ.setdefault("FOD_WGRAD_QUEUE_EAGER", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "future-object-detection_amd"))
import torch
from types import SimpleNamespace
import bench
from future_od.datasets.synthetic import make_batch
from future_od.native import lib as L
from future_od.optim import FusedAdamW

model, detr = bench.build(SimpleNamespace(), "cuda:0", False, 5, "bf16")
model.eval()
opt = FusedAdamW(model.parameters(), lr=detr.lr, weight_decay=detr.weight_decay, max_norm=detr.max_norm)
data = make_batch(2, 6, 900, 1600, seed=1234, device="cuda:0")
seen = collections.Counter()
real_call = L.call


def spy(name, *a, **k):
    if name == "fod_gemm_tn_acc":
        M, N1, K2 = a[7], a[8], a[9]
        site = [f"{f.name}:{f.lineno}" for f in traceback.extract_stack()[:-1] if "functional.py" in f.filename or "backbone.py" in f.filename][-3:]
        seen[(M, N1, K2, " < ".join(reversed(site)))] += 1
    return real_call(name, *a, **k)


for i in range(2):
    if i == 1:
        L.call = spy
        import future_od.native.ops as ops, future_od.native.functional as Fn
        ops.call = spy
        if hasattr(Fn, "L"):
            Fn.L.call = spy
    opt.zero_grad()
    out, _s, loss, stats, od = model(data=data, distributed=False)
    loss.backward()
    opt.step()
torch.cuda.synchronize()
tot = 0
for (M, N1, K2, site), c in sorted(seen.items(), key=lambda kv: -kv[1]):
    print(f"{c:4d} x  M={M:6d} N1={N1:5d} K2={K2:5d}   {site}")
    tot += c
print("direct fod_gemm_tn_acc launches:", tot)
