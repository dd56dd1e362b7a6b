#!/usr/bin/env python3
"""Headline benchmark: frame-sequences/s, forward + backward, T=6, 900x1600, B=2 per GPU, bf16.

    python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run)

One "step" = one pass of the hot path over one synthetic batch: model(data) (backbone, encoder,
decoder, matcher + set loss, post-processing / AP bookkeeping, exactly the reference's forward,
future_od/models/st_detr.py:98-167) + loss.backward() (+ the gradient all-reduce when N > 1) +
gradient clipping and the AdamW update (reference future_od/trainer.py:171-189).  Inputs are resident in
HBM before the timed region.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "future-object-detection_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch                                                        # noqa: E402
import torch.distributed as dist                                    # noqa: E402

PEAK_BF16_TFLOPS = 2500.0      # dense bf16 MFMA, /opt/skills/guides/MI355X_MICROARCH.md "Chip-level parameters"
T_FRAMES, HEIGHT, WIDTH, BATCH_PER_GPU = 6, 900, 1600, 2
# --workload: the headline (default; the configuration BASELINE.json's metric is quoted on) and, for the record in
# DESIGN.md, the other BASELINE.json configs that fit one GPU: (T, H, W, batch / GPU, num_images K, live TFLOP per
# frame-sequence forward + backward from SURVEY.md 8d, or None)
WORKLOADS = {
    "headline": (6, 900, 1600, 2, 5, 4.082e12),
    "headline-k2": (6, 900, 1600, 2, 2, 1.638e12),      # the shipped num_images = 2: 3 of 5 past frames are dead work
    "cfg2": (4, 800, 1333, 2, 2, 1.187e12),             # configs[1]: T=4, 800x1333
    "nusc500-stage1": (3, 448, 800, 4, 2, 0.395e12),    # configs[3], runs/nusc_spatiotemporal_imu_500ms.py: 32 / 8 GPUs
    "nusc500-stage2": (3, 896, 1600, 2, 2, 1.601e12),   # configs[3], second stage: 16 / 8 GPUs
    "t8": (8, 900, 1600, 2, 7, 5.712e12),               # configs[4]'s shape (T=8, all 7 past frames live) in bf16
}
LIVE_FLOPS = WORKLOADS["headline"][5]


MATCHER_EVENTS = []        # measure(): device-matcher failures of a timed run (the loss went non-finite), see there


def live_flops_per_sequence(num_images):
    """Algorithmic FLOPs of one frame-sequence, forward + backward, live work only."""
    return LIVE_FLOPS


def build(args, device, distributed, num_images, dtype="bf16"):
    from types import SimpleNamespace
    from future_od.models.st_detr import SpatioTemporalDETRArgs
    from runs._model import build_model
    torch.manual_seed(0)
    a = SimpleNamespace(device=device, distributed=distributed, compute_dtype=dtype, num_images=num_images,
                        attn_dtype=getattr(args, "attn_dtype", "bf16"))
    detr = SpatioTemporalDETRArgs(num_classes=8, num_queries=128, lr_backbone=1e-4, pretrained_backbone=False)
    return build_model(a, detr), detr


def pmc_traffic_per_launch(entry):
    """HBM bytes per launch of `entry` from the committed PMC passes (profiles/pmc_traffic.json, written by
    tools/pmc_traffic.py from separate --pmc FETCH_SIZE / WRITE_SIZE runs of this same workload; FETCH doubled as
    MI355X_MICROARCH.md prescribes for gfx950).  PMC passes cannot run inside the timed bench; None if absent."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            return json.load(f)[entry]["bytes_per_launch"]
    except (OSError, KeyError, ValueError):
        return None


# device kernel name (as the tracer reports it, demangled or not) -> C-ABI entry point whose launches it serves
_KERNEL_ENTRY = [
    (r"conv2d_fwd_kernel|conv_stem_fwd_kernel|stem_pool_kernel|nt_big_kernel(<|ILi)1|bottleneck_fused", "fod_conv2d_fwd"),
    (r"conv2d_dgrad|nt_big_kernel(<|ILi)[23]", "fod_conv2d_dgrad"),
    (r"tn_big_kernel|tn_reduce_kernel", "fod_conv2d_wgrad_acc"),
    (r"gemm_nt_small_kernel|gemm_nt_kernel|gemm_nt_grouped|nt_big_kernel(<|ILi)0|linear_add_norm|mlp2_mul", "fod_gemm_nt"),
    (r"gemm_tn_multi_long", "fod_gemm_tn_multi_long"), (r"gemm_tn_multi", "fod_gemm_tn_multi"),
    (r"gemm_tn|colsum", "fod_gemm_tn_acc"),
    (r"attn_quant_fp8", "fod_attn_quant_fp8"), (r"attn_fwd_fp8", "fod_attn_fwd_fp8"),
    (r"attn_fwd", "fod_attn_fwd"), (r"attn_bwd", "fod_attn_bwd"),
    (r"ln_fwd", "fod_layernorm_fwd"), (r"ln_bwd", "fod_layernorm_bwd"), (r"eltwise|dropout_kernel", "fod_eltwise"),
    (r"multi_adamw", "fod_multi_adamw"), (r"multi_sqnorm", "fod_multi_sqnorm_det"), (r"multi_permute3", "fod_multi_permute3"),
    (r"maxpool", "fod_maxpool3x3s2"), (r"stem_layout", "fod_clip_to_stem_layout"), (r"lap_dev", "fod_lap_solve_batch_dev"),
    (r"od_map", "fod_od_map"), (r"set_loss", "fod_set_loss"), (r"permute3|nchw", "fod_permute3_cast"),
]


def _entry_of(kernel_name):
    import re
    for pat, entry in _KERNEL_ENTRY:
        if re.search(pat, kernel_name):
            return entry
    return "torch/other"


def replay_kernel_times(replay, replays=5):
    """Device time per C-ABI entry point of ONE replayed step -- the product's launch mode -- from torch.profiler
    (roctracer / rocprofiler-sdk kernel records: begin / end timestamps taken by the GPU).  One tracing session around
    `replays` replays; the tracer comes up asynchronously and misses the first replay(s) (observed on the box: 2 replays
    traced -> one replay's 1001 kernels), so the number of replays actually captured is counted from a kernel that runs
    exactly once per step (fod_post_proc's) and whole replays only are kept.  Returns
    ({entry: {"ms_per_step", "kernels_per_step"}}, kernels per step, replays captured) or None without a tracer."""
    try:
        from torch.autograd import DeviceType
        from torch.profiler import ProfilerActivity, profile
        with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:   # (device-only: no events come back)
            for _ in range(replays):
                replay()
            torch.cuda.synchronize()
        ev = sorted((e for e in prof.events() if e.device_type == DeviceType.CUDA and "emcpy" not in e.name and "emset" not in e.name),
                    key=lambda e: e.time_range.start)
        marks = [i for i, e in enumerate(ev) if "post_proc_kernel" in e.name]
        if len(marks) < 2:
            return None
        # whole steps: from one once-per-step kernel to the next
        ev = ev[marks[0]:marks[-1]]
        captured = len(marks) - 1
        agg = {}
        for e in ev:
            a = agg.setdefault(_entry_of(e.name), [0, 0.0])
            a[0] += 1
            a[1] += e.device_time                          # microseconds
        return ({k: {"ms_per_step": v[1] / captured / 1e3, "kernels_per_step": v[0] / captured} for k, v in agg.items()},
                len(ev) / captured, captured)
    except Exception as exc:                               # noqa: BLE001  (no tracer: the eager event leg is the fallback)
        sys.stderr.write(f"bench: torch.profiler unavailable ({exc!r}); falling back to the eager event leg\n")
        return None


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(num_images, cores, budget_s=150.0):
    """the oracle (plain-PyTorch fp32 restatement of the reference graph, validated against the
    fixtures generated from the reference's own files) on this box's host cores, `model.eval()` math with autograd
    on, the AS-SHIPPED loop (no dead-work skipping: every past frame's backbone / encoder / decoder pass runs, as in
    reference paper.py:347-350), one frame-sequence (B=1) of the bench workload's shape, forward + backward incl. set
    loss: 1 warm-up + 3 timed iterations, median (fewer timed iterations if the budget would be exceeded -- said in
    `sample`).  Plus BASELINE.json configs[0]: ResNet-18, 1+1 layers, (1,6,3,224,224), forward only.
    Checker code used as the baseline leg only; it is never on the product path."""
    import statistics
    from future_od.datasets.synthetic import make_batch
    from oracle import criterion as ocrit
    from oracle import stdetr as O
    torch.set_num_threads(cores)
    cfg = O.Config(num_images=num_images)
    sd = O.make_state_dict(cfg, 0)
    for k, (_, kind) in O.param_spec(cfg).items():
        if kind == "param":
            sd[k].requires_grad_(True)
    data = make_batch(1, T_FRAMES, HEIGHT, WIDTH, seed=1, max_boxes=40)

    def one():
        for v in sd.values():
            v.grad = None
        t0 = time.perf_counter()
        out = O.core_forward(sd, cfg, data["video"], O.imu_from_data(data), skip_dead=False)
        loss, _, _ = ocrit.total_loss(cfg, out, data)
        loss.backward()
        return time.perf_counter() - t0

    warm = one()
    n_timed = 3 if warm * 4 <= budget_s else 1
    times = [one() for _ in range(n_timed)]
    dt = statistics.median(times)
    # configs[0]: the reference's own CPU-runnable case, forward only
    cfg1 = O.Config(backbone="resnet18", enc_layers=1, dec_layers=1)
    sd1 = O.make_state_dict(cfg1, 0)
    d1 = make_batch(1, 6, 224, 224, seed=1, max_boxes=10)
    with torch.no_grad():
        O.core_forward(sd1, cfg1, d1["video"], O.imu_from_data(d1), skip_dead=False)
        t1 = []
        for _ in range(3):
            t0 = time.perf_counter()
            O.core_forward(sd1, cfg1, d1["video"], O.imu_from_data(d1), skip_dead=False)
            t1.append(time.perf_counter() - t0)
    return {"value": 1.0 / dt, "unit": "frame-sequences/s", "cores": cores, "kind": "port", "cpu_model": _cpu_model(),
            "sample": f"1 frame-sequence (B=1, T={T_FRAMES}, {HEIGHT}x{WIDTH}, num_images={num_images}), fp32, eval-mode "
                      f"math with autograd, forward (as-shipped frame loop, no dead-work skipping) + set loss + backward; "
                      f"1 warm-up ({warm:.1f} s) + {n_timed} timed, median {dt:.1f} s",
            "cfg1_forward": {"value": 1.0 / statistics.median(t1), "unit": "frame-sequences/s",
                             "sample": "configs[0]: ResNet-18, 1+1 layers, (1,6,3,224,224), forward only, "
                                       f"1 warm-up + 3 timed, median {1e3 * statistics.median(t1):.0f} ms"}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--num-images", type=int, default=5, help="decoder cross-attention blocks K (5 = all past frames live)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the K=2 / fp32 side measurements")
    ap.add_argument("--no-graph", action="store_true",
                    help="N=1 only: launch every kernel from Python (the default replays the step as one hipGraph)")
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--attn-dtype", default="bf16", choices=["bf16", "fp8", "fp8-all"],
                    help="BASELINE.json configs[4]: fp8 = MX-fp8 (e4m3, block-scaled MFMA) QK^T / PV in the long-sequence "
                         "attention launches (encoder self-attention), fp8-all = in every attention launch; with "
                         "--workload t8 this is configs[4], reported beside the bf16 line of the same workload")
    ap.add_argument("--forward-only", action="store_true",
                    help="the evaluation pass (no_grad forward incl. set loss, post-processing, AP bookkeeping) instead of "
                         "the training step; N = 1; for the record, not the headline")
    ap.add_argument("--train-mode", action="store_true",
                    help="model.train(): the reference's dropout 0.1 active (BASELINE.md measures model.eval() with "
                         "autograd on, the default here); for the record, not the headline")
    ap.add_argument("--rehearse", action="store_true",
                    help="tiny shapes, gloo backend, every rank on cuda:0: exercises the N>1 code path on a 1-GPU box")
    ap.add_argument("--force-ddp", action="store_true",
                    help="with --gpus 1: run the N>1 code path (RCCL process group of one rank, FodDataParallel, "
                         "distributed loss normalisation) to measure its overhead on a 1-GPU box")
    ap.add_argument("--workload", default="headline", choices=sorted(WORKLOADS),
                    help="default: the headline configuration; the others are for the record (DESIGN.md 5)")
    a = ap.parse_args()
    global T_FRAMES, HEIGHT, WIDTH, BATCH_PER_GPU, LIVE_FLOPS
    if a.workload != "headline":
        T_FRAMES, HEIGHT, WIDTH, BATCH_PER_GPU, a.num_images, LIVE_FLOPS = WORKLOADS[a.workload]
    elif a.num_images != 5:
        LIVE_FLOPS = {2: 1.638e12}.get(a.num_images)
    if a.rehearse:
        T_FRAMES, HEIGHT, WIDTH = 4, 128, 192
        import faulthandler                       # a rehearsal that hangs says where
        faulthandler.dump_traceback_later(150, exit=True)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1 or a.force_ddp
    if a.rehearse:
        local = 0
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            os.environ.setdefault("MASTER_PORT", "29533")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        torch.cuda.set_device(local)
        dist.init_process_group(backend="gloo" if a.rehearse else "nccl", init_method="env://")
    assert world == a.gpus or not distributed, (world, a.gpus)
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)

    from future_od.datasets.synthetic import make_batch
    from future_od.native import lib as L
    from future_od.optim import FusedAdamW

    def fence():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    def measure(num_images, dtype, steps, warmup, profile):
        """Build the model, run `warmup` untimed and exactly `steps` timed steps between fences; returns
        (seconds over the timed steps -- max over ranks, final loss, profiler summary or None, profiled steps)."""
        # data parallel + graphs: the bare model (GraphedStep averages the gradients itself between its two graphs);
        # data parallel, eager: the FodDataParallel wrapper with its side-stream reducer
        model, detr = build(a, device, distributed and not use_graph, num_images, dtype)
        model.eval()     # forward + backward in model.eval() with autograd on (dropout off, FrozenBN)
        if a.train_mode:
            model.train()
        opt = FusedAdamW(model.parameters(), lr=detr.lr, weight_decay=detr.weight_decay, max_norm=detr.max_norm)
        data = make_batch(BATCH_PER_GPU, T_FRAMES, HEIGHT, WIDTH, seed=1234 + rank, device=device)

        def step():
            opt.zero_grad()
            out, _state, loss, stats, od = model(data=data, distributed=distributed)
            loss.backward()
            opt.step()
            return loss

        if a.forward_only:
            def step():                                  # noqa: F811  (the evaluation pass of the reference's Trainer)
                with torch.no_grad():
                    out, _state, loss, stats, od = model(data=data, distributed=distributed)
                return loss

        eager_step = step
        if use_graph and a.forward_only:
            from future_od.graph import GraphedForward
            fwd = GraphedForward(model)
            fwd(data)
            step = lambda: fwd(data)[1]
        elif use_graph:
            # the same step -- same kernels, same order -- captured once and replayed (future_od/graph.py); the
            # capture and its eager warm-up steps happen before the timed region
            from future_od.graph import GraphedStep
            graphed = GraphedStep(model, opt, warmup=2, data_parallel=distributed)
            if distributed:
                graphed.broadcast_parameters()
            graphed(data)
            step = lambda: graphed(data)[1]

        for _ in range(warmup):
            step()
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            loss = step()
        fence()
        dt = time.perf_counter() - t0
        t = torch.tensor([dt], device=device)
        if distributed:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        # the device-side matcher reports a non-finite cost matrix one step late through a per-device status word: look at
        # it HERE, so that a run of this model that went non-finite (the random-init model at lr 1e-4 does in some runs,
        # DESIGN.md 5) is reported on ITS line instead of raising inside the next model built in this process
        try:
            from future_od.models.set_criterion import _lap_status
            _lap_status(torch.device(device)).check(wait=True)
        except Exception as e:                               # noqa: BLE001
            MATCHER_EVENTS.append(f"{dtype}{'/' + a.attn_dtype if getattr(a, 'attn_dtype', 'bf16') != 'bf16' else ''}, "
                                  f"num_images={num_images}: {e}")
        summ, nprof, replay = None, 2, None
        if profile and use_graph:
            # the profiling leg launches eagerly but must time the kernels the replayed step is made of: the weight
            # gradients that a captured step queues and launches together are queued here too (an eager step does not
            # by default: it is bound by the launching thread, and the bookkeeping costs it time)
            from future_od.native import functional as Fn
            Fn.WGRADS.eager = True
        if profile and rank != 0:
            for _ in range(nprof + 1):               # the steps hold collectives: every rank runs them, rank 0 measures
                eager_step()
        if profile and rank == 0:
            # (1) FLOPs and call counts per entry point, and -- as a cross-check / fallback -- event-pair times of eagerly
            # launched steps.  The collector is held off and, where the device can park a stream on a host flag, the
            # GPU does not start the step before the launching thread has queued all of it: an event pair then brackets
            # back-to-back device work, not a host stall (tools/probe_roofline_modes.py).
            import ctypes
            import gc
            import threading
            eager_step()
            torch.cuda.synchronize()
            gc.collect()
            gc.disable()
            flag, parked = ctypes.c_void_p(), False
            try:
                if L._ENTRY["fod_stream_wait_supported"](device.index or 0):
                    L._plain_call("fod_host_flag_create", ctypes.addressof(flag))
                    parked = True
            except Exception:                        # noqa: BLE001
                parked = False
            L.PROFILER.start()
            for i in range(nprof):
                if parked:
                    L._plain_call("fod_stream_wait_flag", flag.value, i + 1, torch.cuda.current_stream().cuda_stream)
                    release = threading.Timer(3.0, L._plain_call, ("fod_host_flag_set", flag.value, i + 1))
                    release.start()                  # a full queue must not leave the stream parked for good
                eager_step()
                if parked:
                    L._plain_call("fod_host_flag_set", flag.value, i + 1)
                    release.cancel()
                    torch.cuda.synchronize()
            L.PROFILER.stop()
            summ = L.PROFILER.summary()
            gc.enable()
            if parked:
                L._plain_call("fod_host_flag_destroy", flag.value)
            eager_leg_info.update({"stream_parked_during_enqueue": parked, "gc_disabled": True})
        if profile and use_graph:
            Fn.WGRADS.eager = False
        if profile and use_graph:
            # (2) the times that go into `roofline`: kernel records of the REPLAYED graph (every rank replays -- the
            # data-parallel step holds collectives -- rank 0 traces)
            if rank == 0:
                replay = replay_kernel_times(step)
            else:
                for _ in range(5):
                    step()
        final = float(loss.detach())
        if distributed and profile:
            # what the first real multi-GPU run needs to be diagnosable: who took part, how many gradient bytes were
            # averaged and how, and how much of the communication was NOT hidden behind the backbone's backward
            seen = torch.ones(1, device=device)
            dist.all_reduce(seen)
            st = dict(model.grad_reducer.stats) if not use_graph else {}
            nbytes = 4 * sum(p.numel() for p in getattr(model, "module", model).parameters() if p.requires_grad)
            fence()
            t1 = time.perf_counter()
            for _ in range(3):
                if use_graph:
                    graphed(data, sync=False)        # same graphs, no gradient average (the LAST thing this model does)
                else:
                    opt.zero_grad()
                    with model.no_sync():
                        _o, _s, l2, _st, _od = model(data=data, distributed=distributed)
                        l2.backward()
                    opt.step()
            fence()
            t_ns = torch.tensor([(time.perf_counter() - t1) / 3], device=device)
            dist.all_reduce(t_ns, op=dist.ReduceOp.MAX)
            ddp_info.update({"n_ranks_seen": int(seen.item()), "backend": dist.get_backend(),
                             "grad_allreduce_bytes_per_step": nbytes,
                             "ms_per_step_without_comm": 1e3 * float(t_ns.item()),
                             "exposed_comm_ms_per_step": 1e3 * (dt / steps - float(t_ns.item()))})
            if use_graph:
                ddp_info.update({"mode": "graph A (forward + backward) -> eager all-reduce -> graph B (clip + AdamW)",
                                 "allreduce_tensors": graphed.comm_stats["tensors"],
                                 "allreduce_bytes": graphed.comm_stats["bytes"]})
            else:
                ddp_info.update({"mode": "eager launches, arena regions all-reduced in place on a side stream during "
                                         "the backbone's backward",
                                 "arena_flushes": st["arena_flushes"], "arena_bytes_in_place": 4 * st["arena_elems"],
                                 "straggler_tensors": st["stragglers"]})
        del model, opt, data
        import gc
        gc.collect()
        torch.cuda.empty_cache()
        return dt, final, summ, nprof, replay

    # N = 1: the whole step is one hipGraph.  N > 1: two graphs (forward + backward, clip + AdamW) with the gradient
    # all-reduce launched eagerly between them (future_od/graph.py); --no-graph: every kernel launched from Python with
    # the all-reduces overlapped with the backbone's backward (future_od/parallel.py)
    use_graph = not a.no_graph
    ddp_info, eager_leg_info = {}, {}
    dt, final_loss, summ, nprof, replay = measure(a.num_images, a.dtype, a.steps, a.warmup, not a.no_roofline)
    seqs = BATCH_PER_GPU * world * a.steps
    value = seqs / dt

    result = {
        "metric": f"frame-sequences/sec fwd+bwd, T={T_FRAMES} {HEIGHT}x{WIDTH}", "value": value,
        "unit": "frame-sequences/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * dt / a.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
        "config": {"workload": f"spatiotemporal ConditionalDETR (ResNet-50, 6 enc + 6 dec layers, 128 queries), "
                               f"T={T_FRAMES} frames {HEIGHT}x{WIDTH}, batch {BATCH_PER_GPU}/GPU, "
                               f"num_images K={a.num_images} ({a.num_images} past frames live), random-init weights; "
                               f"step = forward (incl. Hungarian set loss, post-proc, AP bookkeeping) + backward + "
                               f"clip + AdamW",
                   "global_batch": BATCH_PER_GPU * world, "frames": T_FRAMES, "resolution": [HEIGHT, WIDTH],
                   "num_images": a.num_images, "parallelism": f"dp{world}"},
        "final_loss": final_loss,
        "model_mode": "train (dropout 0.1 active)" if a.train_mode else "eval (BASELINE.md: eval-mode math, autograd on)",
        "step": "evaluation pass (no_grad forward + set loss + post-processing)" if a.forward_only else "training step",
        "launch_mode": ("eager (one Python call per kernel)" if not use_graph else
                        "hipgraph replay (one graph per step)" if not distributed else
                        "hipgraph replay (forward + backward graph, eager gradient all-reduce, optimizer graph)"),
    }
    # what the number rests on: the step was replayed from a captured graph (a failed capture raises
    # in GraphedStep -- bench.py has no eager fallback), and whether the weight-gradient queues were available (they need a
    # private torch symbol; without it ~170 extra launches run one by one)
    from future_od.native import functional as _Fn
    if MATCHER_EVENTS:
        result["matcher_events"] = list(MATCHER_EVENTS)
    result["fast_paths"] = {"captured_graph": bool(use_graph), "wgrad_queue": bool(_Fn.WGRADS.enabled),
                            "wgrad_queue_long": bool(_Fn.WGRADS.enabled and _Fn.WGRADS.long_enabled)}
    if use_graph and not _Fn.WGRADS.enabled and rank == 0:
        sys.stderr.write("bench: the weight-gradient queue is OFF (torch._C._current_graph_task_id missing or FOD_WGRAD_QUEUE=0): "
                         "every Linear weight gradient is its own launch\n")
    if ddp_info:
        result["ddp"] = ddp_info
    fl = live_flops_per_sequence(a.num_images)
    if fl:
        result["model_tflops_per_gpu"] = fl * BATCH_PER_GPU * a.steps / dt / 1e12
        result["model_frac_of_bf16_peak"] = result["model_tflops_per_gpu"] / PEAK_BF16_TFLOPS

    if rank == 0 and summ is not None:
        # per entry point and step: algorithmic FLOPs and calls (from the eagerly launched steps) and device time -- from
        # the replayed graph's kernel records when the tracer is there, else from the event pairs of the eager leg
        work = {k: v["work"] / nprof for k, v in summ.items()}
        calls = {k: v["calls"] / nprof for k, v in summ.items()}
        if replay is not None:
            times = {k: v["ms_per_step"] for k, v in replay[0].items()}
            nlaunch = {k: v["kernels_per_step"] for k, v in replay[0].items()}
            timing = (f"kernel records (torch.profiler / roctracer) of the replayed hipGraph, mean over {replay[2]} whole "
                      "replays; FLOPs and call counts from the same step launched eagerly")
            result["kernels_per_replayed_step"] = replay[1]
        else:
            times = {k: 1e3 * v["seconds"] / nprof for k, v in summ.items()}
            nlaunch = dict(calls)
            timing = "HIP events around every C-ABI call of eagerly launched steps (collector off, stream parked during enqueue)"
        total = sum(times.values())
        step_ms = 1e3 * dt / a.steps
        result["device_ms_per_step_profiled"] = total
        result["fod_launches_per_step"] = sum(calls.values())
        # clip + optimizer reported separately (they ARE inside `ms_per_step`, which is therefore
        # conservative): device time of the gradient-norm and AdamW launches
        result["clip_adamw_ms_per_step"] = sum(times.get(k, 0.0) for k in ("fod_multi_sqnorm_acc", "fod_multi_sqnorm_det", "fod_multi_adamw"))
        top = sorted(times.items(), key=lambda kv: -kv[1])
        name, ms = top[0]
        outliers = {k: {"calls": v["outliers"], "median_us": v["median_us"], "max_us": v["max_us"]}
                    for k, v in summ.items() if v["outliers"]}
        eager_total = sum(1e3 * v["seconds"] / nprof for v in summ.values())
        result["eager_event_leg"] = dict(eager_leg_info, device_ms_per_step=eager_total, calls_over_10x_median=outliers)
        # a roofline whose kernel time does not fit inside the step it was taken from is not evidence: say so instead
        # (the event-pair fallback brackets each launch with its ~2.5 us dispatch gap: the gate allows for them)
        allowance = 1.1 * step_ms + (0.0 if replay is not None else 3e-3 * sum(calls.values()))
        if total > allowance or ms > step_ms or not work.get(name):
            result["roofline_invalid"] = {"reason": "profiled device time does not fit the timed step" if work.get(name)
                                          else "dominant entry has no FLOP count", "kernel": name, "kernel_ms_per_step": ms,
                                          "sum_entries_ms_per_step": total, "ms_per_step": step_ms, "timing": timing}
        else:
            achieved = work[name] / (ms * 1e-3) / 1e12
            result["roofline"] = {"bound": "mfma", "kernel": name, "achieved": achieved, "peak": PEAK_BF16_TFLOPS,
                                  "unit": "TFLOP/s", "frac": achieved / PEAK_BF16_TFLOPS,
                                  "traffic": pmc_traffic_per_launch(name),
                                  "avg_launch_us": 1e3 * ms / max(nlaunch.get(name, 1.0), 1.0),
                                  "launches_per_step": nlaunch.get(name), "ms_per_step": ms,
                                  "share_of_device_time": ms / total, "timing": timing,
                                  "check": {"sum_entries_ms_per_step": total, "ms_per_step": step_ms}}
        result["kernel_breakdown"] = {k: {"ms_per_step": v, "launches_per_step": nlaunch.get(k), "calls_per_step": calls.get(k),
                                          "tflops": (work[k] / (v * 1e-3) / 1e12) if work.get(k) and v > 0 else None}
                                      for k, v in top[:14]}
    # for the record (N = 1, headline only): the reference's AS-SHIPPED model (num_images = 2: 3 of 5 past frames are
    # dead work and skipped) and the fp32-MFMA parity mode, same step definition, driver-timed like the main line
    if world == 1 and a.workload == "headline" and a.num_images == 5 and not a.no_extras and not a.rehearse:
        extras = {}
        for key, (k_img, dtp) in {"headline-k2": (2, a.dtype), "headline-fp32": (5, "fp32")}.items():
            if dtp == a.dtype and k_img == a.num_images:
                continue
            st = 4
            edt, eloss, _, _, _ = measure(k_img, dtp, st, 2, False)
            extras[key] = {"value": BATCH_PER_GPU * st / edt, "unit": "frame-sequences/s", "ms_per_step": 1e3 * edt / st,
                           "steps": st, "warmup": 2, "num_images": k_img, "dtype": dtp, "final_loss": eloss}
        result["also"] = extras
    if a.attn_dtype != "bf16":
        # BASELINE.json configs[4] asks for fp8 "vs bf16": the same workload with the bf16 attention kernels, same process,
        # same box, and the attention entries' device time per replayed step for both
        result["dtype"] = f"{a.dtype} (attention QK^T / PV: MX-fp8 e4m3, {a.attn_dtype})"
        attn_keys = ("fod_attn_fwd_fp8", "fod_attn_quant_fp8", "fod_attn_fwd", "fod_attn_bwd")
        mine = {k: v for k, v in (replay[0].items() if replay else []) if k in attn_keys}
        fp8_mode, a.attn_dtype = a.attn_dtype, "bf16"
        bdt, bloss, _bs, _bn, breplay = measure(a.num_images, a.dtype, a.steps, a.warmup, not a.no_roofline)
        a.attn_dtype = fp8_mode
        theirs = {k: v for k, v in (breplay[0].items() if breplay else []) if k in attn_keys}
        result["vs_bf16_attention"] = {
            "bf16": {"value": BATCH_PER_GPU * world * a.steps / bdt, "ms_per_step": 1e3 * bdt / a.steps, "final_loss": bloss,
                     "attention_entries": theirs},
            "fp8": {"value": value, "ms_per_step": 1e3 * dt / a.steps, "final_loss": final_loss, "attention_entries": mine},
            "throughput_ratio_fp8_over_bf16": value / (BATCH_PER_GPU * world * a.steps / bdt)}
    if distributed:
        dist.barrier()
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(a.num_images, min(os.cpu_count() or 1, 64))
    if rank == 0:
        print(json.dumps(result))
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
