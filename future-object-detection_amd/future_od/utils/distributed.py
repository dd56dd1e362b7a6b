"""Process-group setup and the small logging collectives (reference future_od/utils/distributed.py).

One process per GPU; backend "nccl" is RCCL on ROCm (xGMI inside a node).  The gradient all-reduce
itself lives in future_od.parallel (bucketed, overlapped with backward); what is here are the
per-iteration scalars, batched into as few collectives as the data allows."""
import os
import signal
import threading

import torch
import torch.distributed as distrib

EXIT = threading.Event()
EXIT.clear()


def _clean_exit_handler(signum, frame):
    EXIT.set()
    print("Exiting cleanly", flush=True)


def install_signal_handlers():
    for sig in (signal.SIGINT, signal.SIGTERM, signal.SIGUSR2):
        try:
            signal.signal(sig, _clean_exit_handler)
        except ValueError:        # not the main thread
            pass


install_signal_handlers()


def disable_prints_unless_master(is_master):
    import builtins
    builtin_print = builtins.print

    def print(*args, **kwargs):
        force = kwargs.pop("force", False)
        if is_master or force:
            builtin_print(*args, **kwargs)

    builtins.print = print


def init_distributed_and_device_(args):
    if args.distributed:
        args.world_size = int(os.environ.get("WORLD_SIZE", os.environ.get("SLURM_NTASKS", 1)))
        args.world_rank = int(os.environ.get("RANK", os.environ.get("SLURM_PROCID", 0)))
        local = int(os.environ.get("LOCAL_RANK", getattr(args, "local_rank", 0)))
        args.local_rank = local
        backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            args.device = torch.device("cuda", local)
            torch.cuda.set_device(args.device)
        else:
            args.device = torch.device("cpu")
        distrib.init_process_group(backend=backend, init_method="env://")
        disable_prints_unless_master(args.world_rank == 0)
    else:
        args.local_rank = 0
        args.world_rank = 0
        args.world_size = 1


def reduce_distrib_loss(input_dict, average=True):
    """All-reduce a dict of scalar tensors in ONE collective (keys sorted for cross-rank consistency)."""
    world_size = distrib.get_world_size()
    with torch.no_grad():
        names = sorted(input_dict.keys())
        values = torch.stack([input_dict[k].detach().float().reshape(()) for k in names], dim=0)
        distrib.all_reduce(values)
        if average:
            values /= world_size
    return {k: v for k, v in zip(names, values)}


def gather_distrib_od_map_stuffs(inputs):
    """All-gather the four od_map tensors; bool tensors travel as uint8."""
    world_size = distrib.get_world_size()
    out = []
    with torch.no_grad():
        for t in inputs:
            src = t.to(torch.uint8) if t.dtype == torch.bool else t
            bufs = [torch.zeros_like(src) for _ in range(world_size)]
            distrib.all_gather(bufs, src.contiguous())
            out.append([b.to(t.dtype) for b in bufs])
    return out
