"""Which in-process per-kernel timing survives on the box?  (round 3)

  a) eager leg with events, as round 2 shipped it                     (host stalls land inside event pairs)
  b) the same leg with the stream PARKED on a host flag while the whole step is enqueued, then released
     (the GPU never waits for the launching thread: event pairs bracket back-to-back device work)
  c) torch.profiler (roctracer) over graph replays: kernel durations of the product's launch mode

Prints per-entry totals for each and the step time of the replayed graph.
"""
import ctypes as C
import gc
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "future-object-detection_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)
import torch  # noqa: E402

import bench  # noqa: E402
from future_od.datasets.synthetic import make_batch  # noqa: E402
from future_od.graph import GraphedStep  # noqa: E402
from future_od.native import functional as Fn  # noqa: E402
from future_od.native import lib as L  # noqa: E402
from future_od.optim import FusedAdamW  # noqa: E402


def main():
    small = len(sys.argv) > 1 and sys.argv[1] == "small"
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    a = type("A", (), {})()
    model, detr = bench.build(a, dev, False, 5, "bf16")
    model.eval()
    opt = FusedAdamW(model.parameters(), lr=detr.lr, weight_decay=detr.weight_decay, max_norm=detr.max_norm)
    T, H, W = (6, 900, 1600) if not small else (3, 128, 192)
    data = make_batch(2, T, H, W, seed=1234, device=dev)

    def eager_step():
        opt.zero_grad()
        out, _s, loss, stats, od = model(data=data, distributed=False)
        loss.backward()
        opt.step()
        return loss

    graphed = GraphedStep(model, opt, warmup=2, data_parallel=False)
    graphed(data)
    for _ in range(3):
        graphed(data)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        graphed(data)
    torch.cuda.synchronize()
    ms_graph = (time.perf_counter() - t0) * 100
    print(f"graph replay: {ms_graph:.3f} ms/step", flush=True)

    Fn.WGRADS.eager = True
    eager_step()
    torch.cuda.synchronize()

    def summarize(tag, recs):
        agg = {}
        for name, work, e0, e1 in recs:
            agg.setdefault(name, []).append(e0.elapsed_time(e1) * 1e3)
        tot = 0.0
        rows = []
        for k, v in agg.items():
            v.sort()
            med = v[len(v) // 2]
            rows.append((sum(v), k, len(v), med, v[-1], sum(1 for x in v if x > 10 * med)))
            tot += sum(v)
        rows.sort(reverse=True)
        print(f"--- {tag}: total {tot / 1e3:.3f} ms over {len(recs)} calls")
        for s, k, n, med, mx, nout in rows[:14]:
            print(f"  {s / 1e3:8.3f} ms  {n:4d} calls  median {med:8.1f} us  max {mx:9.1f} us  >10x median: {nout}  {k}")
        sys.stdout.flush()
        return tot / 1e3

    # a) plain eager leg
    gc.collect()
    L.PROFILER.start()
    eager_step()
    L.PROFILER.stop()
    torch.cuda.synchronize()
    summarize("a) eager + events (round-2 method)", L.PROFILER.records)

    # a2) eager leg, gc disabled
    gc.collect()
    gc.disable()
    L.PROFILER.start()
    eager_step()
    L.PROFILER.stop()
    torch.cuda.synchronize()
    gc.enable()
    summarize("a2) eager + events, gc disabled", L.PROFILER.records)

    # b) parked stream
    flag = C.c_void_p()
    L._plain_call("fod_host_flag_create", C.addressof(flag))
    stream = torch.cuda.current_stream().cuda_stream
    for rep in range(2):
        ticket = rep + 1
        gc.collect()
        gc.disable()
        torch.cuda.synchronize()
        L._plain_call("fod_stream_wait_flag", flag.value, ticket, stream)
        timer = threading.Timer(5.0, lambda: L._plain_call("fod_host_flag_set", flag.value, ticket))   # never leave it parked
        timer.start()
        t0 = time.perf_counter()
        L.PROFILER.start()
        eager_step()
        L.PROFILER.stop()
        t_enq = time.perf_counter() - t0
        L._plain_call("fod_host_flag_set", flag.value, ticket)
        timer.cancel()
        torch.cuda.synchronize()
        gc.enable()
        print(f"parked leg {rep}: enqueue took {t_enq * 1e3:.1f} ms (a value near 5000 = the queue filled and the timer released it)")
        summarize("b) stream parked during enqueue", L.PROFILER.records)
    Fn.WGRADS.eager = False

    # c) torch.profiler over replays
    try:
        from torch.profiler import ProfilerActivity, profile
        with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
            for _ in range(2):
                graphed(data)
            torch.cuda.synchronize()
        ev = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
        agg = {}
        for e in ev:
            a_ = agg.setdefault(e.name, [0, 0.0])
            a_[0] += 1
            a_[1] += e.device_time
        rows = sorted(((v[1], k, v[0]) for k, v in agg.items()), reverse=True)
        print(f"--- c) torch.profiler over 2 replays: {len(ev)} device events, total {sum(r[0] for r in rows) / 2e3:.3f} ms/step")
        for s, k, n in rows[:16]:
            print(f"  {s / 2e3:8.3f} ms/step  {n // 2:4d}/step  {k[:100]}")
    except Exception as exc:  # noqa: BLE001
        print("torch.profiler failed:", repr(exc))


if __name__ == "__main__":
    main()
