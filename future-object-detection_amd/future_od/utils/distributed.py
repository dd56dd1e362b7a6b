"""Process-group setup and the small logging collectives (reference future_od/utils/distributed.py).

One process per GPU; backend "nccl" is RCCL on ROCm (xGMI inside a node).  The gradient all-reduce
itself lives in future_od.parallel (bucketed, overlapped with backward); what is here are the
per-iteration scalars, batched into as few collectives as the data allows."""
import builtins
import os
import signal
import threading

import torch
import torch.distributed as distrib

# Set by SIGINT / SIGTERM / SIGUSR2 (slurm's pre-emption notice): the Trainer finishes the iteration it is in,
# then returns, so a checkpoint is never cut off in the middle of a write.
EXIT = threading.Event()


def _request_exit(signum, _frame):
    EXIT.set()
    builtins.print(f"signal {signum}: finishing the current iteration, then exiting", flush=True)


def install_signal_handlers(signals=(signal.SIGINT, signal.SIGTERM, signal.SIGUSR2)):
    """Route the given signals to EXIT.  Only the main thread may install handlers; elsewhere this is a no-op."""
    if threading.current_thread() is not threading.main_thread():
        return False
    for s in signals:
        signal.signal(s, _request_exit)
    return True


install_signal_handlers()


class _RankFilteredPrint:
    """print() for one-process-per-GPU runs: only the master rank's lines reach stdout unless force=True is given."""

    def __init__(self, is_master):
        self.is_master = bool(is_master)
        self.inner = builtins.print.inner if isinstance(builtins.print, _RankFilteredPrint) else builtins.print

    def __call__(self, *values, force=False, **kw):
        if self.is_master or force:
            self.inner(*values, **kw)


def disable_prints_unless_master(is_master):
    builtins.print = _RankFilteredPrint(is_master)


def _env_int(*names, default=0):
    for n in names:
        if n in os.environ:
            return int(os.environ[n])
    return int(default)


def init_distributed_and_device_(args):
    """Fill args.world_size / world_rank / local_rank / device and, if args.distributed, join the process group given
    by the launcher's environment (torch.distributed.run or slurm).  RCCL ("nccl") when a GPU is visible, gloo
    otherwise (CPU tests)."""
    if not getattr(args, "distributed", False):
        args.world_size, args.world_rank, args.local_rank = 1, 0, 0
        return
    args.world_size = _env_int("WORLD_SIZE", "SLURM_NTASKS", default=1)
    args.world_rank = _env_int("RANK", "SLURM_PROCID", default=0)
    args.local_rank = _env_int("LOCAL_RANK", default=getattr(args, "local_rank", 0))
    on_gpu = torch.cuda.is_available()
    args.device = torch.device("cuda", args.local_rank) if on_gpu else torch.device("cpu")
    if on_gpu:
        torch.cuda.set_device(args.device)
    distrib.init_process_group(backend="nccl" if on_gpu else "gloo", init_method="env://")
    disable_prints_unless_master(args.world_rank == 0)


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
