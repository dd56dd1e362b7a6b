"""Where does the split backward (FOD_GRAPH_OVERLAP) lose its time at ONE rank?  One process, a one-rank RCCL group,
the captured data-parallel step in five variants, interleaved rounds (same device, same clocks):
   plain        graph A -> all-reduce -> graph B                                   (the round-2 default)
   plain-nocomm the same without the all-reduce
   split-async  A1 -> [transformer all-reduce on RCCL's stream] || A2 -> backbone all-reduce -> B
   split-after  A1 -> A2 -> both all-reduces -> B                                  (the split's own cost + sequential comm)
   split-nocomm A1 -> A2 -> B
    python tools/ddp_overlap_probe.py [rounds=3] [steps=10]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "future-object-detection_amd")]
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29544")
os.environ.setdefault("RANK", "0")
os.environ.setdefault("WORLD_SIZE", "1")
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import bench  # noqa: E402
from future_od.datasets.synthetic import make_batch  # noqa: E402
from future_od.graph import GraphedStep  # noqa: E402
from future_od.optim import FusedAdamW  # noqa: E402


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", init_method="env://")
    from types import SimpleNamespace
    data = make_batch(2, bench.T_FRAMES, bench.HEIGHT, bench.WIDTH, seed=1234, device=dev)
    # ONE model and optimizer, three captured steps of it (the prepared-operand registry serves one live model)
    model, detr = bench.build(SimpleNamespace(), dev, False, 5, "bf16")
    model.eval()
    opt = FusedAdamW(model.parameters(), lr=detr.lr, weight_decay=detr.weight_decay, max_norm=detr.max_norm)
    variants = {}
    for name, overlap, mode in (("plain", False, "async"), ("split-async", True, "async"), ("split-after", True, "after")):
        g = GraphedStep(model, opt, warmup=2, data_parallel=True)
        g.overlap, g.overlap_mode = overlap, mode
        g.broadcast_parameters()
        g(data)
        variants[name] = (g, True)
        if name != "split-after":
            variants[name.split("-")[0] + "-nocomm"] = (g, False)
    res = {k: [] for k in variants}
    for _ in range(rounds):
        for name, (g, sync) in variants.items():
            for _ in range(3):
                g(data, sync=sync)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                g(data, sync=sync)
            torch.cuda.synchronize()
            res[name].append(1e3 * (time.perf_counter() - t0) / steps)
    for name, v in res.items():
        print(f"{name:13s} ms/step: " + "  ".join(f"{x:7.3f}" for x in v) + f"   (min {min(v):.3f})")
    dist.destroy_process_group()


main()
