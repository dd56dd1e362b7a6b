"""Two gloo ranks on one GPU: how long does an all-reduce of a large CUDA tensor take -- fresh allocation, an offset slice
of a bigger buffer, several smaller tensors?  (The data-parallel rehearsal's 228 MB all-reduce took seconds.)"""
import os, sys, time
import torch, torch.distributed as dist
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo")
dev = torch.device("cuda", 0)
big = torch.zeros(80_000_000, device=dev)


def t(label, fn, reps=3):
    fn(); torch.cuda.synchronize(); dist.barrier()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    if rank == 0:
        print(f"{label:44s} {dt * 1e3:9.1f} ms", flush=True)


fresh = torch.zeros(57_000_000, device=dev)
t("57 M floats, own allocation", lambda: dist.all_reduce(fresh))
sl = big[64 * 1000: 64 * 1000 + 57_000_000]
t("57 M floats, slice at a 256-byte offset", lambda: dist.all_reduce(sl))
parts = [torch.zeros(19_000_000, device=dev) for _ in range(3)]
t("3 x 19 M floats", lambda: [dist.all_reduce(p) for p in parts])
small = torch.zeros(1_000_000, device=dev)
t("1 M floats", lambda: dist.all_reduce(small))
cpu = torch.zeros(57_000_000)
t("57 M floats on the host", lambda: dist.all_reduce(cpu))
dist.destroy_process_group()
