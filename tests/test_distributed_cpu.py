"""N>1 host logic on CPU with gloo, world_size 2: batch sharding, the num_boxes all-reduce that scales
every loss term, the one-collective stat reduction, od_map gathers (bools travel as uint8)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from future_od.datasets.synthetic import make_batch
    from future_od.models.set_criterion import SetCriterion
    from future_od.models.st_detr import to_detr_targets
    from future_od.utils.distributed import gather_distrib_od_map_stuffs, reduce_distrib_loss
    data = make_batch(2, 3, 32, 48, seed=1234 + rank, max_boxes=9)          # per-rank shard of the global batch
    targets = to_detr_targets(32, 48, data["active"], data["boxes"], data["classes"])
    crit = SetCriterion(8, matcher=None, weight_dict={}, focal_alpha=0.25, losses=["labels", "boxes", "cardinality"],
                        matching_mode="per level")
    local = float(sum(len(t["labels"]) for t in targets))
    nb = crit.global_num_boxes(targets, torch.device("cpu"), distributed=True)
    stats = reduce_distrib_loss({"b": torch.tensor(float(rank + 1)), "a": torch.tensor(10.0 * (rank + 1))})
    od = (torch.full((2, 3, 4), float(rank)), torch.ones(2, 3, 4, dtype=torch.bool) if rank else torch.zeros(2, 3, 4, dtype=torch.bool),
          torch.zeros(3, 4, 4, dtype=torch.bool), torch.full((3, 4), rank, dtype=torch.int64))
    g = gather_distrib_od_map_stuffs(od)
    q.put((rank, local, nb, {k: float(v) for k, v in stats.items()},
           [float(t.float().mean()) for t in g[0]], [t.dtype == torch.bool for t in g[1]], [int(t[0, 0]) for t in g[3]]))
    dist.barrier()
    dist.destroy_process_group()


def test_world_size_two_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    locals_ = [r[1] for r in res]
    assert locals_[0] != locals_[1]                       # different shards
    for r in res:
        assert r[2] == pytest.approx(max(sum(locals_) / world, 1.0))      # mean boxes per rank, same on all ranks
        assert r[3] == {"a": pytest.approx(15.0), "b": pytest.approx(1.5)}
        assert r[4] == [0.0, 1.0] and r[5] == [True, True] and r[6] == [0, 1]


def _ddp_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from future_od.parallel import FodDataParallel
    torch.manual_seed(0)                                   # same initial weights; rank 0's are broadcast anyway
    net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.ReLU(), torch.nn.Linear(5, 3))
    ddp = FodDataParallel(net)
    g = torch.Generator().manual_seed(100 + rank)
    x = torch.randn(4, 6, generator=g)
    # reference: local gradients without communication, then averaged by hand
    with ddp.no_sync():
        ddp(x).pow(2).sum().backward()
    local = [p.grad.clone() for p in net.parameters()]
    want = []
    for t in local:
        bufs = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(bufs, t)
        want.append(sum(bufs) / world)
    for p in net.parameters():
        p.grad = None
    ddp(x).pow(2).sum().backward()                         # the product path: gradients averaged when backward returns
    got = [p.grad.clone() for p in net.parameters()]
    err = max(float((a - b).abs().max()) for a, b in zip(got, want))
    differ = max(float((a - b).abs().max()) for a, b in zip(local, want))
    q.put((rank, err, differ, dict(ddp.grad_reducer.stats)))
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_reducer_averages_over_two_gloo_ranks():
    """FodDataParallel's own reducer (torch's is bypassed) on CPU tensors: gradients that do not live in the device
    arena travel as one packed all-reduce per dtype and come back averaged, exactly as hand-averaged no_sync()
    gradients; isinstance(DistributedDataParallel) and .module keep working for the reference's Trainer."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_ddp_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, err, differ, stats in res:
        assert err < 1e-6, (rank, err)
        assert differ > 1e-3                                # the ranks really had different gradients
        assert stats["stragglers"] == 4 and stats["arena_flushes"] == 0
