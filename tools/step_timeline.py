"""GPU and host timeline of the training step without a profiler: CUDA events at the phase boundaries of several
steady-state steps, host clocks at the same points.  GPU time between two events minus the kernel time there is
idle time (host-bound stretches, the matcher bubble)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "future-object-detection_amd"))
import torch
import bench
from future_od.datasets.synthetic import make_batch
from future_od.optim import FusedAdamW
from future_od.models import set_criterion as SC
from types import SimpleNamespace

a = SimpleNamespace(dtype="bf16")
dev = torch.device("cuda", 0)
DDP = os.environ.get("FORCE_DDP") == "1"
if DDP:
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29534")
    torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl", init_method="env://", rank=0, world_size=1)
model, detr = bench.build(a, dev, DDP, 5, "bf16")
if DDP and os.environ.get("NOOP_COMM") == "1":
    from torch.distributed.algorithms.ddp_comm_hooks.debugging_hooks import noop_hook
    model.register_comm_hook(None, noop_hook)
model.eval()      # the benchmark graph: eval mode with autograd on
opt = FusedAdamW(model.parameters(), lr=1e-4, weight_decay=1e-4, max_norm=0.1)
data = make_batch(2, 6, 900, 1600, seed=1234, device=dev)

marks = []          # (label, event, host time)


def mark(label):
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    marks.append((label, e, time.perf_counter()))


orig_match = SC.HungarianMatcher.match_levels


def match_levels(self, logits, boxes, packed, threads=8):
    mark("matcher in")
    r = orig_match(self, logits, boxes, packed, threads)
    mark("matcher out")
    return r


SC.HungarianMatcher.match_levels = match_levels


def step():
    mark("step start")
    opt.zero_grad()
    out, _s, loss, stats, od = model(data=data, distributed=DDP)
    mark("forward queued")
    loss.backward()
    mark("backward queued")
    opt.step()
    mark("optimizer queued")


for _ in range(4):
    step()
torch.cuda.synchronize()
marks.clear()
N = 8
t0 = time.perf_counter()
for _ in range(N):
    step()
mark("end")
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / N
print(f"wall {1e3*wall:.2f} ms/step")
agg = {}
for (la, ea, ha), (lb, eb, hb) in zip(marks, marks[1:]):
    k = f"{la} -> {lb}"
    g, h = agg.setdefault(k, [0.0, 0.0])
    agg[k] = [g + ea.elapsed_time(eb), h + 1e3 * (hb - ha)]
print(f"{'segment':44s} {'GPU ms':>8s} {'host ms':>8s}   (per step)")
for k, (g, h) in agg.items():
    print(f"{k:44s} {g/N:8.2f} {h/N:8.2f}")
# how far the host runs ahead: host time of a mark vs the time its event completed
base_h = marks[0][2]
lag = {}
for label, e, h in marks:
    gpu_t = marks[0][1].elapsed_time(e)           # ms after the first event on the GPU timeline
    lag.setdefault(label, []).append(gpu_t - 1e3 * (h - base_h))
print("GPU completion minus host issue time (ms; grows when the GPU is the bottleneck, ~0 when the host is):")
for k, v in lag.items():
    print(f"  {k:20s} " + " ".join(f"{x:6.1f}" for x in v[:8]))
