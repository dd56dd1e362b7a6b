"""Plain write / copy rates on one MI355X for the sizes of the encoder's GEMM outputs (torch fill_ / copy_)."""
import time, torch


def t(fn, it=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(it): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / it * 1e6


for mb in (7.4, 14.8, 29.7, 59.4, 118.8, 475.0):
    n = int(mb * 1e6 / 2)
    x = torch.empty(n, device="cuda", dtype=torch.bfloat16); y = torch.empty_like(x)
    tf = min(t(lambda: x.fill_(1.0)) for _ in range(3))
    tc = min(t(lambda: y.copy_(x)) for _ in range(3))
    print(f"{mb:6.1f} MB: fill {tf:6.1f} us = {mb / tf:5.2f} TB/s   copy {tc:6.1f} us = {2 * mb / tc:5.2f} TB/s (read + write)")
