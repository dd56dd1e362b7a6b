"""Is the divergence of the random-init headline model at lr 1e-4 the
kernels' doing or the optimisation's?  One GPU, evidence in three parts:

  1. run the captured bf16 step on rotating synthetic batches, snapshotting parameters + AdamW moments every 5 steps,
     until the gradient norm takes off (> 30 x its median over the first steps) or a loss goes non-finite;
  2. at the snapshot taken BEFORE the take-off: every parameter gradient of one frame-sequence from (a) the bf16 kernels,
     (b) the fp32-MFMA parity mode, (c) the CPU oracle (plain PyTorch fp32, oracle/) -- if (b) agrees with (c) to parity
     tolerance and (a) with both to bf16 tolerance, the gradients that feed the blow-up are CORRECT;
  3. from the same snapshot (parameters, moments, step count) continue in the fp32 parity mode on the same batches: if
     that run takes off too, it is the workload (lr 1e-4 on a random-init ResNet-50 with identity frozen BatchNorms),
     not bf16 rounding, atomics order, the key-split hand-offs or the queued weight gradients.

    python tools/divergence_control.py [max_trials=6] [steps=200]        (writes a report to stdout)
"""
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "future-object-detection_amd")]
import torch  # noqa: E402

import bench  # noqa: E402
from future_od.datasets.synthetic import make_batch  # noqa: E402
from future_od.graph import GraphedStep  # noqa: E402
from future_od.optim import FusedAdamW  # noqa: E402

DEV = torch.device("cuda", 0)
T, H, W = bench.T_FRAMES, bench.HEIGHT, bench.WIDTH


def build(dtype):
    from types import SimpleNamespace
    model, detr = bench.build(SimpleNamespace(), DEV, False, 5, dtype)
    model.eval()
    opt = FusedAdamW(model.parameters(), lr=detr.lr, weight_decay=detr.weight_decay, max_norm=detr.max_norm)
    return model, opt


def snapshot(model, opt):
    ps = [p.detach().clone() for p in model.parameters()]
    ms = [(opt.state[p]["exp_avg"].clone(), opt.state[p]["exp_avg_sq"].clone()) if p in opt.state and "exp_avg" in opt.state[p]
          else None for p in model.parameters()]
    return ps, ms, getattr(opt, "_step_no", 0)


def restore(model, opt, snap):
    ps, ms, step_no = snap
    with torch.no_grad():
        for p, s, m in zip(model.parameters(), ps, ms):
            p.copy_(s)
            if m is not None:
                st = opt._state_for(p)
                st["exp_avg"].copy_(m[0])
                st["exp_avg_sq"].copy_(m[1])
    opt._step_no = step_no
    if getattr(opt, "_dev_step", None) is not None:
        opt._dev_step.fill_(float(step_no))
    from future_od.native import functional as Fn
    Fn.PREP.mark_stale()


def run_until_takeoff(model, opt, batches, steps, every=5):
    """Returns (took_off, trajectory [(step, loss, grad norm)], snapshots [(step, snap)])."""
    step = GraphedStep(model, opt, warmup=2)
    traj, snaps = [], []
    for i in range(steps):
        if i % every == 0:
            snaps.append((i, snapshot(model, opt)))
            snaps = snaps[-4:]
        out = step(batches[i % len(batches)])
        gn = float(opt._sq[0]) ** 0.5
        loss = float(out[1])
        traj.append((i, loss, gn))
        base = statistics.median(t[2] for t in traj[:20]) if len(traj) >= 10 else None
        if loss != loss or abs(loss) == float("inf") or gn != gn or (base and gn > 30.0 * base):
            # the gradients of THIS replay are still in place: which tensors are non-finite, and where inside them?
            rep = []
            for n_, p_ in model.named_parameters():
                if p_.grad is None:
                    continue
                g_ = p_.grad.detach()
                bad_ = ~torch.isfinite(g_)
                nb = int(bad_.sum())
                if nb:
                    idx = bad_.reshape(-1).nonzero().flatten()
                    rep.append((n_, nb, g_.numel(), tuple(g_.shape), idx[:6].tolist(), idx[-3:].tolist(),
                                float(g_[torch.isfinite(g_)].abs().max()) if nb < g_.numel() else float("nan")))
            big = sorted(((float(p_.grad.detach().float().abs().nan_to_num(0, 0, 0).max()), n_) for n_, p_ in model.named_parameters()
                          if p_.grad is not None), reverse=True)[:5]
            print(f"   at the flagged replay: {len(rep)} gradient tensors hold non-finite values")
            for r_ in rep[:12]:
                print(f"      {r_[0]} {r_[3]}: {r_[1]} of {r_[2]} non-finite, first flat indices {r_[4]}, last {r_[5]}, largest finite |g| {r_[6]:.3e}")
            print("   largest finite |g| entries:", [(f"{v:.3e}", n_) for v, n_ in big], flush=True)
            return True, traj, snaps
    return False, traj, snaps


def gradients(model, data):
    for p in model.parameters():
        p.grad = None
    _, _, loss, _, _ = model(data=data, distributed=False)
    loss.backward()
    torch.cuda.synchronize()
    return float(loss), {n: p.grad.detach().float().cpu() for n, p in model.named_parameters() if p.grad is not None}


def main():
    trials = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    batches = [make_batch(bench.BATCH_PER_GPU, T, H, W, seed=50 + i, device=DEV) for i in range(4)]
    found = None
    for trial in range(trials):
        model, opt = build("bf16")
        t0 = time.perf_counter()
        off, traj, snaps = run_until_takeoff(model, opt, batches, steps)
        print(f"trial {trial}: bf16, lr 1e-4, {len(traj)} steps in {time.perf_counter() - t0:.1f} s: "
              f"{'TAKE-OFF at step %d' % traj[-1][0] if off else 'no take-off'}; gradient norm first / median / last: "
              f"{traj[0][2]:.3e} / {statistics.median(t[2] for t in traj):.3e} / {traj[-1][2]:.3e}", flush=True)
        if off:
            found = (model, opt, traj, snaps)
            break
        del model, opt
        torch.cuda.empty_cache()
    if found is None:
        print(f"no run took off in {trials} trials of {steps} steps: nothing to examine")
        return
    model, opt, traj, snaps = found
    print("last 12 steps (step, loss, gradient norm):")
    for t in traj[-12:]:
        print(f"   {t[0]:4d}  {t[1]:12.4f}  {t[2]:.4e}")
    # the snapshot at least 5 steps before the take-off
    t_off = traj[-1][0]
    s_step, snap = [s for s in snaps if s[0] <= t_off - 5][-1] if any(s[0] <= t_off - 5 for s in snaps) else snaps[0]
    print(f"examining the state before step {s_step} (take-off flagged at step {t_off})")
    data = batches[s_step % len(batches)]
    one = {k: (v[:1].contiguous() if isinstance(v, torch.Tensor) else v) for k, v in data.items() if k != "_host_annotations"}

    # ---- 1b. the same steps again from the snapshot, launched eagerly (bf16): which gradient tensors go non-finite first?
    restore(model, opt, snap)
    opt.disable_device_step() if hasattr(opt, "disable_device_step") else None
    for k in range(14):
        data_k = batches[(s_step + k) % len(batches)]
        opt.zero_grad()
        _, _, loss_k, _, _ = model(data=data_k, distributed=False)
        loss_k.backward()
        torch.cuda.synchronize()
        named = [(n_, p_.grad) for n_, p_ in model.named_parameters() if p_.grad is not None]
        norms = torch.stack(torch._foreach_norm([g_ for _, g_ in named])).float().cpu()
        bad = [named[i][0] for i in range(len(named)) if not bool(torch.isfinite(norms[i]))]
        fin = norms.nan_to_num(0.0, 0.0, 0.0)
        top = int(fin.argmax())
        print(f"   eager bf16 step {s_step + k}: loss {float(loss_k):.4f}, gradient norm {float((fin ** 2).sum() ** 0.5):.4e} (finite part), "
              f"largest {float(fin[top]):.3e} at {named[top][0]}; {len(bad)} non-finite tensors{': ' + ', '.join(bad[:5]) if bad else ''}", flush=True)
        if bad:
            break
        opt.step()

    # ---- 2. gradients at the snapshot: bf16 kernels, fp32 kernels, CPU oracle
    restore(model, opt, snap)
    l16, g16 = gradients(model, one)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    del model, opt, found
    torch.cuda.empty_cache()
    m32, o32 = build("fp32")
    m32.load_state_dict(sd)
    l32, g32 = gradients(m32, one)
    from oracle import criterion as ocrit
    from oracle import stdetr as O
    torch.set_num_threads(min(os.cpu_count() or 1, 64))
    cfg = O.Config(num_images=5)
    osd = {k: v.detach().float().cpu().clone() for k, v in sd.items()}
    for k, (_, kind) in O.param_spec(cfg).items():
        if kind == "param":
            osd[k].requires_grad_(True)
    one_cpu = {k: (v.cpu() if isinstance(v, torch.Tensor) else v) for k, v in one.items()}
    t0 = time.perf_counter()
    ref = O.core_forward(osd, cfg, one_cpu["video"], O.imu_from_data(one_cpu), skip_dead=True)
    lo, _, _ = ocrit.total_loss(cfg, ref, one_cpu)
    lo.backward()
    print(f"oracle forward + backward: {time.perf_counter() - t0:.1f} s; loss oracle {float(lo):.6f}, fp32 kernels {l32:.6f}, "
          f"bf16 kernels {l16:.6f}")
    go = {k: v.grad for k, v in osd.items() if getattr(v, "grad", None) is not None}
    rows, tot = [], {"o": 0.0, "e32": 0.0, "e16": 0.0, "32": 0.0, "16": 0.0}
    for name, r in go.items():
        if name not in g32:
            continue
        rn, e32, e16 = float(r.norm()), float((g32[name] - r).norm()), float((g16[name] - r).norm())
        rows.append((name, rn, e32, e16))
        tot["o"] += rn ** 2
        tot["e32"] += e32 ** 2
        tot["e16"] += e16 ** 2
        tot["32"] += float(g32[name].norm()) ** 2
        tot["16"] += float(g16[name].norm()) ** 2
    gl = tot["o"] ** 0.5
    print(f"{len(rows)} gradient tensors compared.  global gradient norm: oracle {gl:.4e}, fp32 kernels {tot['32'] ** 0.5:.4e}, "
          f"bf16 kernels {tot['16'] ** 0.5:.4e}")
    print(f"whole gradient: ||g_fp32 - g_oracle|| / ||g_oracle|| = {tot['e32'] ** 0.5 / gl:.3e};  ||g_bf16 - g_oracle|| / ||g_oracle|| = {tot['e16'] ** 0.5 / gl:.3e}")
    print("largest absolute deviations of the fp32 kernels (tensor, ||g_ref||, ||g32 - g_ref||, ||g16 - g_ref||):")
    for name, rn, e32, e16 in sorted(rows, key=lambda t: -t[2])[:8]:
        print(f"   {name:70s} {rn:.3e} {e32:.3e} {e16:.3e}")
    print("largest absolute deviations of the bf16 kernels:")
    for name, rn, e32, e16 in sorted(rows, key=lambda t: -t[3])[:8]:
        print(f"   {name:70s} {rn:.3e} {e32:.3e} {e16:.3e}")

    # ---- 3. continue from the snapshot in the fp32 parity mode
    restore(m32, o32, snap)
    off32, traj32, _ = run_until_takeoff(m32, o32, batches[s_step % 4:] + batches[:s_step % 4], 60)
    print(f"fp32 parity mode continued from the same parameters / moments / step count for {len(traj32)} steps: "
          f"{'TAKE-OFF at +%d steps' % traj32[-1][0] if off32 else 'no take-off'}; gradient norm first / last: "
          f"{traj32[0][2]:.3e} / {traj32[-1][2]:.3e}")
    for t in traj32[:4] + traj32[-6:]:
        print(f"   +{t[0]:3d}  {t[1]:12.4f}  {t[2]:.4e}")


main()
