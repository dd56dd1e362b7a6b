"""Debug aid: per-parameter gradient agreement between the HIP model and the CPU oracle."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "future-object-detection_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from oracle import stdetr as O, criterion as ocrit
from future_od.datasets.synthetic import make_batch
from test_model_gpu import build_product, CASES

name = sys.argv[1] if len(sys.argv) > 1 else "g5_r50_2x2"
dtype = {"fp32": torch.float32, "bf16": torch.bfloat16}[sys.argv[2] if len(sys.argv) > 2 else "fp32"]
cfg = CASES[name]
B, L, H, W, seed = {"g5_r50_2x2": (2, 4, 96, 160, 12), "g5_cfg1_r18": (1, 6, 224, 224, 11), "g5_r18_k3_noimu": (2, 5, 64, 96, 13)}[name]
model, sd = build_product(cfg, dtype, seed)
data_cpu = make_batch(B, L, H, W, seed=seed, max_boxes=12)
data = {k: (v.to("cuda:0") if isinstance(v, torch.Tensor) else v) for k, v in data_cpu.items()}
out, _, loss, stats, od = model(data=data, distributed=False)
loss.backward()
spec = O.param_spec(cfg)
for k, (_, kind) in spec.items():
    if kind == "param":
        sd[k].requires_grad_(True)
o = O.core_forward(sd, cfg, data_cpu["video"], O.imu_from_data(data_cpu) if cfg.use_imu else None)
l2, _, _ = ocrit.total_loss(cfg, o, data_cpu)
l2.backward()
print("loss", float(loss), float(l2))
rows = []
for n, p in model.named_parameters():
    if not p.requires_grad:
        continue
    g = p.grad.float().cpu().flatten() if p.grad is not None else None
    r = sd[n].grad.flatten() if sd[n].grad is not None else None
    if g is None or r is None:
        rows.append((9.0, n, "none", g is None, r is None)); continue
    rn, gn = float(r.norm()), float(g.norm())
    cos = float(torch.dot(g, r) / (gn * rn + 1e-30)) if rn > 0 and gn > 0 else 1.0
    rows.append((1 - cos, n, f"cos={cos:.5f} ratio={gn / (rn + 1e-30):.4f} refnorm={rn:.3e}"))
for r in rows:
    print(*r[1:])
rows.sort(reverse=True)
print("...", len(rows), "params; worst 1-cos =", rows[0][0])
