"""Where the HOST time of a step goes (cProfile over a few steps of a small workload, where the step is host-bound:
the GPU work of BASELINE configs[1] / configs[3] is shorter than the ~21 ms the host needs to issue ~1400 launches)."""
import cProfile, os, pstats, sys, io
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "future-object-detection_amd"))
import torch
from types import SimpleNamespace
import bench
from future_od.datasets.synthetic import make_batch
from future_od.optim import FusedAdamW

dev = torch.device("cuda", 0)
T, H, W, B, K, _ = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "nusc500-stage1"]
model, detr = bench.build(SimpleNamespace(), dev, False, K, "bf16")
model.eval()
opt = FusedAdamW(model.parameters(), lr=1e-4, weight_decay=1e-4, max_norm=0.1)
data = make_batch(B, T, H, W, seed=1234, device=dev)


def step():
    opt.zero_grad()
    out, _s, loss, stats, od = model(data=data, distributed=False)
    loss.backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    step()
torch.cuda.synchronize()
pr.disable()
for key in ("tottime", "cumulative"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(45)
    print(s.getvalue()[:9000])
