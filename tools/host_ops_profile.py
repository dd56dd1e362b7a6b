"""Host (CPU) time of one step by operator, both threads (torch.profiler, CPU activity): which autograd Functions and
aten ops the launching threads spend their ~26 ms in."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "future-object-detection_amd"))
import torch
from torch.profiler import profile, ProfilerActivity
from types import SimpleNamespace
import bench
from future_od.datasets.synthetic import make_batch
from future_od.optim import FusedAdamW

dev = torch.device("cuda", 0)
T, H, W, B, K, _ = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "headline"]
model, detr = bench.build(SimpleNamespace(), dev, False, K, "bf16")
model.eval()
opt = FusedAdamW(model.parameters(), lr=1e-4, weight_decay=1e-4, max_norm=0.1)
data = make_batch(B, T, H, W, seed=1234, device=dev)


def step():
    opt.zero_grad()
    out, _s, loss, stats, od = model(data=data, distributed=False)
    loss.backward()
    opt.step()


for _ in range(4):
    step()
torch.cuda.synchronize()
n = 3
with profile(activities=[ProfilerActivity.CPU]) as prof:
    for _ in range(n):
        step()
    torch.cuda.synchronize()
ka = prof.key_averages()
tot = sum(e.self_cpu_time_total for e in ka)
print(f"self CPU time total {tot / n / 1e3:.2f} ms per step (profiler overhead included)")
for e in sorted(ka, key=lambda e: -e.self_cpu_time_total)[:45]:
    print(f"{e.self_cpu_time_total / n / 1e3:8.3f} ms self  {e.cpu_time_total / n / 1e3:8.3f} ms total  {e.count / n:7.1f} calls  {e.key[:70]}")
