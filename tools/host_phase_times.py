"""Host-side duration of each phase of a step (forward call, backward call, optimizer) next to the step time: a phase
whose host time is close to its GPU time is stalling the launching thread.  usage: host_phase_times.py [ddp]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "future-object-detection_amd"))
import torch
import torch.distributed as dist
from types import SimpleNamespace
import bench
from future_od.datasets.synthetic import make_batch
from future_od.optim import FusedAdamW

ddp = len(sys.argv) > 1 and sys.argv[1] == "ddp"
if os.environ.get("NOGC"):
    import gc
    gc.disable()
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
if ddp:
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
    os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
    dist.init_process_group(backend="nccl", init_method="env://")
model, detr = bench.build(SimpleNamespace(), dev, ddp, 5, "bf16")
model.eval()
opt = FusedAdamW(model.parameters(), lr=1e-4, weight_decay=1e-4, max_norm=0.1)
data = make_batch(2, 6, 900, 1600, seed=1234, device=dev)
acc = {"fwd": 0.0, "bwd": 0.0, "opt": 0.0}


def step(rec=False):
    t0 = time.perf_counter()
    opt.zero_grad()
    out, _s, loss, stats, od = model(data=data, distributed=ddp)
    t1 = time.perf_counter()
    loss.backward()
    t2 = time.perf_counter()
    opt.step()
    t3 = time.perf_counter()
    if rec:
        acc["fwd"] += t1 - t0; acc["bwd"] += t2 - t1; acc["opt"] += t3 - t2


for _ in range(4):
    step()
torch.cuda.synchronize()
n = 6
t0 = time.perf_counter()
for _ in range(n):
    step(True)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"{'ddp' if ddp else 'single'}: {1e3 * dt / n:.2f} ms/step; host time in forward call {1e3 * acc['fwd'] / n:.2f}, "
      f"backward call {1e3 * acc['bwd'] / n:.2f}, optimizer {1e3 * acc['opt'] / n:.2f} ms")
if ddp:
    dist.destroy_process_group()
