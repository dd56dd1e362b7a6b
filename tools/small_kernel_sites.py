"""Which host call sites launch the small torch kernels (fills, copies, adds, muls) of one training step.
Runs torch.profiler with stacks over one step and groups device kernels by the innermost frame that lies in
this repository."""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "future-object-detection_amd"))
import torch
from torch.profiler import profile, ProfilerActivity
import bench
from future_od.datasets.synthetic import make_batch
from future_od.optim import FusedAdamW
from types import SimpleNamespace

a = SimpleNamespace(dtype="bf16")
dev = torch.device("cuda", 0)
model, detr = bench.build(a, dev, False, 5, "bf16")
model.eval()      # the benchmark graph: eval mode with autograd on
opt = FusedAdamW(model.parameters(), lr=1e-4, weight_decay=1e-4, max_norm=0.1)
data = make_batch(2, 6, 900, 1600, seed=1234, device=dev)


def step():
    opt.zero_grad()
    out, _s, loss, stats, od = model(data=data, distributed=False)
    loss.backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()

PAT = ("FillFunctor", "copyBuffer", "CUDAFunctor_add", "MulFunctor", "CUDAFunctorOnSelf", "CatArray", "fillBuffer",
       "direct_copy", "elementwise_kernel", "reduce_kernel", "rocprim", "index")
sites = collections.defaultdict(lambda: [0, 0.0])
kinds = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.events():
    if ev.device_type is None:
        continue
    kt = sum(k.duration for k in ev.kernels) if ev.kernels else 0.0
    if not ev.kernels:
        continue
    names = [k.name for k in ev.kernels]
    if not any(any(p in n for p in PAT) for n in names):
        continue
    site = "?"
    for fr in (ev.stack or []):
        if "/repo/" in fr or "future_od" in fr or "bench.py" in fr:
            site = fr.split("/")[-1] if "/" in fr else fr
            break
    key = (ev.name, site)
    sites[key][0] += len(ev.kernels); sites[key][1] += kt
    for k in ev.kernels:
        short = k.name.split("<")[0][-40:] + ("<" + k.name.split("<")[2][:40] if k.name.count("<") > 1 else "")
        kinds[short][0] += 1; kinds[short][1] += k.duration
print("== by kernel kind")
for k, (n, t) in sorted(kinds.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"{n:5d} {t:9.1f} us  {k}")
print("== by (aten op, repo frame)")
for k, (n, t) in sorted(sites.items(), key=lambda kv: -kv[1][1])[:70]:
    print(f"{n:5d} {t:9.1f} us  {k[0]:28s} {k[1]}")
