"""cProfile (cumulative) of the host side of the training step on the N>1 code path with a single-rank RCCL group."""
import cProfile, io, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "future-object-detection_amd"))
import torch
import torch.distributed as dist
import bench
from future_od.datasets.synthetic import make_batch
from future_od.optim import FusedAdamW
from types import SimpleNamespace
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29535")
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", init_method="env://", rank=0, world_size=1)
a = SimpleNamespace(dtype="bf16")
dev = torch.device("cuda", 0)
model, detr = bench.build(a, dev, True, 5, "bf16")
model.eval()      # the benchmark graph: eval mode with autograd on
opt = FusedAdamW(model.parameters(), lr=1e-4, weight_decay=1e-4, max_norm=0.1)
data = make_batch(2, 6, 900, 1600, seed=1234, device=dev)
def step():
    opt.zero_grad()
    out, _s, loss, stats, od = model(data=data, distributed=True)
    loss.backward()
    opt.step()
for _ in range(4):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(3):
    step()
torch.cuda.synchronize()
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45); print(s.getvalue()[:9000])
dist.destroy_process_group()
