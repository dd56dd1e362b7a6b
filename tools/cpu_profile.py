"""cProfile of the host side of one training step (GPU work is asynchronous, so cumulative time here is
launch + Python overhead, except at the matcher's sync)."""
import cProfile, io, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "future-object-detection_amd"))
import torch
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
# host time of fwd and bwd separately (sync before each phase so the host never waits on the queue)
opt.zero_grad(); torch.cuda.synchronize(); t0 = time.perf_counter()
out, _s, loss, stats, od = model(data=data, distributed=False)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
loss.backward(); t3 = time.perf_counter(); torch.cuda.synchronize(); t4 = time.perf_counter()
opt.step(); t5 = time.perf_counter(); torch.cuda.synchronize()
print(f"host: forward queued in {1e3*(t1-t0):.1f} ms (+{1e3*(t2-t1):.1f} drain), backward queued in {1e3*(t3-t2):.1f} ms (+{1e3*(t4-t3):.1f} drain), optimizer {1e3*(t5-t4):.1f} ms")
pr = cProfile.Profile(); pr.enable()
for _ in range(2):
    step()
torch.cuda.synchronize()
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28); print(s.getvalue()[:6000])
