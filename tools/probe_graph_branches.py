"""Does a replayed hipGraph run independent branches concurrently?  Two chains of tiny kernels (each far from filling
the chip) captured on the capture stream and on a forked side stream, against the same work captured as one chain."""
import torch, time
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
x = [torch.randn(64, 64, device=dev) for _ in range(2)]
w = torch.randn(64, 64, device=dev) * 0.1


def chain(t, n):
    for _ in range(n):
        t = torch.tanh(t @ w)
    return t


def bench(g, reps=20):
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


N = 200
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    chain(x[0], 3); chain(x[1], 3)
torch.cuda.synchronize()
g1 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g1):
    a = chain(x[0], N)
    b = chain(x[1], N)
g2 = torch.cuda.CUDAGraph()
side = torch.cuda.Stream()
with torch.cuda.graph(g2):
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    with torch.cuda.stream(side):
        b2 = chain(x[1], N)
    a2 = chain(x[0], N)
    main.wait_stream(side)
g3 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g3):
    a3 = chain(x[0], N)
print(f"one chain of {2 * N} kernel pairs: {bench(g1):8.1f} us")
print(f"two forked chains of {N}:          {bench(g2):8.1f} us")
print(f"a single chain of {N}:             {bench(g3):8.1f} us")
print("results equal:", torch.equal(a, a2), torch.equal(b, b2))
