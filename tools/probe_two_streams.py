"""Timing probe: do two captured whole-step graphs of HALF the batch, replayed on two streams at once, finish sooner than
one graph of the whole batch?  (A replayed graph runs its kernels one after another; ~6 ms of the headline step are
launches of a few workgroups each -- the decoder's query side -- that leave the chip idle.  Two micro-batches in flight
let one sample's short launches run beside the other's full-chip convolutions.)

TIMING ONLY: the two models of this probe share the gradient arena and the kernels' scratch buffers, so their numbers
are garbage; the product form (future_od/graph.py: micro-batch streams) gives each stream its own."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "future-object-detection_amd"))
import torch
import bench
from types import SimpleNamespace
from future_od.datasets.synthetic import make_batch
from future_od.optim import FusedAdamW
from future_od.graph import GraphedStep

dev = torch.device("cuda", 0)
a = SimpleNamespace(dtype="bf16")


def make(batch, seed):
    model, detr = bench.build(a, dev, False, 5, "bf16")
    model.eval()
    opt = FusedAdamW(model.parameters(), lr=detr.lr, weight_decay=detr.weight_decay, max_norm=detr.max_norm)
    data = make_batch(batch, 6, 900, 1600, seed=seed, device=dev)
    g = GraphedStep(model, opt, warmup=2)
    g(data)
    return g, data


def timeit(fn, n=20, w=5):
    for _ in range(w):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


whole, dw = make(2, 1234)
print(f"one graph, batch 2: {timeit(lambda: whole(dw)):.2f} ms", flush=True)
h0, d0 = make(1, 1234)
h1, d1 = make(1, 1235)
print(f"one graph, batch 1: {timeit(lambda: h0(d0)):.2f} ms", flush=True)


def seq():
    h0(d0); h1(d1)


s0, s1 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)


def conc():
    cur = torch.cuda.current_stream(dev)
    s0.wait_stream(cur); s1.wait_stream(cur)
    with torch.cuda.stream(s0):
        h0(d0)
    with torch.cuda.stream(s1):
        h1(d1)
    cur.wait_stream(s0); cur.wait_stream(s1)


print(f"two graphs of batch 1, one stream: {timeit(seq):.2f} ms", flush=True)
print(f"two graphs of batch 1, two streams: {timeit(conc):.2f} ms", flush=True)
for stagger in (0.25, 0.5):
    # the second stream starts when the first is part-way through its step (busy-wait kernel on the second stream)
    def conc_staggered(frac=stagger):
        cur = torch.cuda.current_stream(dev)
        s0.wait_stream(cur); s1.wait_stream(cur)
        with torch.cuda.stream(s1):
            torch.cuda._sleep(int(frac * 12e-3 * 2.4e9))
            h1(d1)
        with torch.cuda.stream(s0):
            h0(d0)
        cur.wait_stream(s0); cur.wait_stream(s1)
    print(f"  ... second stream delayed by ~{stagger * 12:.0f} ms: {timeit(conc_staggered):.2f} ms", flush=True)
