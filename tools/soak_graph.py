"""Sustained run of the captured headline step on changing batches: loss stays finite and falls, device memory and step
time stay flat, and an eagerly launched step at the end agrees with a replay at the same parameters (the queued weight
gradients, prepared operands and optimizer state all took part in every replay).
    python tools/soak_graph.py [steps=300]
SOAK_LR=1e-5: the random-init model (identity frozen BatchNorms) diverges at the benchmark's 1e-4 in about a third of
the runs after 20-180 steps (gradient norm climbing to inf over several steps) -- an optimisation property
of the synthetic setting; SOAK_PROBE=1 records per step, on the device, which gradient tensor goes non-finite first.
SOAK_BOUND=n, SOAK_SAME_BATCH=1: how far the launching thread may run ahead / one resident batch."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "future-object-detection_amd")]
import torch

import bench
from future_od.datasets.synthetic import make_batch
from future_od.graph import GraphedStep
from future_od.optim import FusedAdamW


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    dev = torch.device("cuda", 0)
    from types import SimpleNamespace
    model, detr = bench.build(SimpleNamespace(), dev, False, 5, "bf16")
    model.eval()
    lr = float(os.environ.get("SOAK_LR", detr.lr))
    opt = FusedAdamW(model.parameters(), lr=lr, weight_decay=detr.weight_decay, max_norm=detr.max_norm)
    batches = [make_batch(bench.BATCH_PER_GPU, bench.T_FRAMES, bench.HEIGHT, bench.WIDTH, seed=50 + i, device=dev)
               for i in range(4)]
    step = GraphedStep(model, opt, warmup=2)
    step(batches[0])
    torch.cuda.synchronize()
    mem0, replays0 = torch.cuda.memory_reserved(), step.replays
    losses, t0, marks = [], time.perf_counter(), []
    hist = torch.zeros(steps, device=dev)
    probe = torch.zeros(steps, 6, device=dev) if os.environ.get("SOAK_PROBE") else None
    names = dict(model.named_parameters())
    w_probe = names["_model.detector.class_embed.weight"]
    g_probe = w_probe.grad
    gnames = [n for n, p in model.named_parameters() if p.grad is not None]
    grads = [p.grad for n, p in model.named_parameters() if p.grad is not None]
    gbad = torch.zeros(steps, 4, device=dev)
    bound = int(os.environ.get("SOAK_BOUND", "0"))     # > 0: the launching thread stays at most that many steps ahead
    events = []
    for i in range(steps):
        if bound and len(events) >= bound:
            events[-bound].synchronize()
        out = step(batches[0 if os.environ.get("SOAK_SAME_BATCH") else i % 4])
        if bound:
            ev = torch.cuda.Event()
            ev.record()
            events.append(ev)
        hist[i].copy_(out[1].detach().float(), non_blocking=True)      # no host sync: the launching thread runs ahead
        if probe is not None:
            post, _l, stats, od = out
            probe[i, 0] = torch.isfinite(post["class_scores"]).all()
            probe[i, 1] = torch.isfinite(post["boxes"]).all()
            probe[i, 2] = opt._sq[0]
            probe[i, 3] = torch.stack([v.float().reshape(-1)[0] for v in stats.values()]).sum()
            probe[i, 4] = w_probe.detach().float().abs().sum()
            probe[i, 5] = g_probe.float().abs().sum() if g_probe is not None else 0
            norms = torch.stack(torch._foreach_norm(grads))
            badn = ~torch.isfinite(norms)
            gbad[i, 0] = badn.sum()
            gbad[i, 1] = badn.float().argmax()
            gbad[i, 2] = norms.nan_to_num(0.0, 0.0, 0.0).max()
            gbad[i, 3] = norms.nan_to_num(0.0, 0.0, 0.0).argmax()
        if os.environ.get("SOAK_EVERY") and not bool(torch.isfinite(out[1])):
            print(f"first non-finite loss at step {i + 1}", flush=True)
            bad = [n for n, p in model.named_parameters() if not bool(torch.isfinite(p).all())]
            print(f"{len(bad)} parameters non-finite, e.g. {bad[:6]}", flush=True)
            badg = [n for n, p in model.named_parameters() if p.grad is not None and not bool(torch.isfinite(p.grad).all())]
            print(f"{len(badg)} gradients non-finite, e.g. {badg[:6]}", flush=True)
            sys.exit(3)
        if i % 25 == 24 or i == steps - 1:
            torch.cuda.synchronize()
            losses.append(float(out[1]))
            marks.append((i + 1, time.perf_counter() - t0))
            print(f"step {i + 1:4d}  loss {losses[-1]:9.4f}  {1e3 * marks[-1][1] / (i + 1):7.2f} ms/step  "
                  f"reserved {torch.cuda.memory_reserved() / 2**30:6.2f} GiB", flush=True)
    bad = (~torch.isfinite(hist)).nonzero().flatten().tolist()
    if bad:
        print(f"non-finite losses from step {bad[0] + 1} on ({len(bad)} of {steps}); before: {hist[max(0, bad[0] - 3):bad[0]].tolist()}")
        if probe is not None:
            lo = max(0, bad[0] - 2)
            print("step: loss | scores finite, boxes finite, grad sqnorm, stats sum, |class_embed.w|, |class_embed.w.grad|")
            lo = max(0, bad[0] - 4)
            for r in range(lo, min(steps, bad[0] + 2)):
                nb, first, mx, amx = gbad[r].tolist()
                print(f"  {r + 1}: {float(hist[r]):.4f} | {probe[r].tolist()} | {int(nb)} non-finite gradient tensors, first "
                      f"{gnames[int(first)] if nb else '-'}; largest finite norm {mx:.3e} at {gnames[int(amx)]}")
        badp = [n for n, p in model.named_parameters() if not bool(torch.isfinite(p).all())]
        print(f"{len(badp)} parameters non-finite, e.g. {badp[:8]}", flush=True)
    assert all(l == l and abs(l) < 1e4 for l in losses), losses
    assert losses[-1] < losses[0], (losses[0], losses[-1])
    assert torch.cuda.memory_reserved() <= mem0 * 1.02 + (64 << 20), (mem0, torch.cuda.memory_reserved())
    assert step.replays == replays0 + steps and len(step._graphs) == 1, (step.replays, len(step._graphs))
    # same parameters, same batch: an eager forward + loss against the next replay's
    with torch.no_grad():
        _, _, l_eager, _, _ = model(data=batches[0], distributed=False)
    l_replay = step(batches[0])[1]
    torch.cuda.synchronize()
    rel = abs(float(l_eager) - float(l_replay)) / max(abs(float(l_eager)), 1.0)
    print(f"eager loss {float(l_eager):.5f} vs replayed {float(l_replay):.5f} (rel {rel:.2e}); replays {step.replays}")
    assert rel < 5e-3
    print("soak ok")


main()
