"""The captured step (future_od/graph.py: one hipGraph for zero_grad + forward + device-side matching + loss +
backward + clip + AdamW) against the eager step: same losses, same parameters, step after step."""
from types import SimpleNamespace

import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def _build(dtype, seed=3):
    from future_od.models.st_detr import SpatioTemporalDETRArgs
    from future_od.optim import FusedAdamW
    from oracle import stdetr as O
    from runs._model import build_model
    cfg = O.Config(backbone="resnet18", enc_layers=1, dec_layers=2)
    args = SimpleNamespace(device=DEV, distributed=False, compute_dtype=dtype, backbone="resnet18")
    detr = SpatioTemporalDETRArgs(num_classes=8, num_queries=128, lr_backbone=1e-4, enc_layers=1, dec_layers=2,
                                  pretrained_backbone=False)
    model = build_model(args, detr)
    model.load_state_dict(O.make_state_dict(cfg, seed))
    model.eval()                                  # eval-mode math with autograd on (the bench's and BASELINE.md's mode)
    opt = FusedAdamW(model.parameters(), lr=1e-4, weight_decay=1e-4, max_norm=0.1)
    return model, opt


def _eager_step(model, opt, data):
    opt.zero_grad()
    post, _, loss, stats, od = model(data=data, distributed=False)
    loss.backward()
    opt.step()
    return loss.detach().clone()


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_graphed_steps_track_eager_steps(dtype):
    from future_od.datasets.synthetic import make_batch
    from future_od.graph import GraphedStep
    data = make_batch(2, 3, 96, 128, seed=11, max_boxes=9, device=DEV)
    data2 = make_batch(2, 3, 96, 128, seed=12, max_boxes=20, device=DEV)      # other targets, same shapes: no re-capture
    m_e, o_e = _build(dtype)
    m_g, o_g = _build(dtype)
    step = GraphedStep(m_g, o_g, warmup=2)
    seq = [data, data, data, data, data2, data, data2]
    le = [float(_eager_step(m_e, o_e, d)) for d in seq]
    # first call = 2 warm-up + 1 device-step eager steps + 1 replay: four optimizer steps, all on `data`
    lg = [None, None, None, float(step(seq[3])[1].detach())]
    lg += [float(step(d)[1].detach()) for d in seq[4:]]
    assert step.replays == 4 and len(step._graphs) == 1
    assert o_g._step_no == o_e._step_no == len(seq)
    tol = 2e-5 if dtype == "fp32" else 2e-3
    for i in range(3, len(seq)):
        assert abs(lg[i] - le[i]) <= tol * max(abs(le[i]), 1.0), (i, lg[i], le[i])
    # parameters after seven steps: a systematic divergence (a skipped or doubled update, stale operands) would show
    # as ~lr = 1e-4 per step; what remains is accumulation-order noise of the f32 atomics amplified by Adam's
    # normalisation on near-zero gradients (a fraction of one lr step)
    worst, total, count = 0.0, 0.0, 0
    for (n, pe), (_, pg) in zip(m_e.named_parameters(), m_g.named_parameters()):
        d = (pe.detach() - pg.detach()).abs()
        worst = max(worst, float(d.max()))
        total += float(d.sum())
        count += d.numel()
    if dtype == "fp32":
        assert worst < 3e-5, worst
    else:
        # bf16 activations: rounding turns the accumulation-order noise into sign flips of near-zero gradients, so
        # single elements can drift by a few lr steps; a systematic difference would move EVERY element by ~1e-4
        assert total / count < 2e-5 and worst < 1.5e-3, (total / count, worst)
    # an eager forward after replays sees the replayed parameters (prepared operands are refreshed)
    with torch.no_grad():
        _, _, l1, _, _ = m_g(data=data, distributed=False)
        _, _, l2, _, _ = m_e(data=data, distributed=False)
    assert abs(float(l1) - float(l2)) <= tol * max(abs(float(l2)), 1.0)


def test_capture_without_spare_job_tables_launches_the_queued_gradients_one_by_one(monkeypatch):
    """The queued short weight gradients (native/functional.py: WGRADS) need a pinned job table set aside BEFORE a
    capture; a capture that finds none (more flushes than spares, a foreign capture) must launch them one by one --
    never allocate inside the capture, never drop a gradient."""
    from future_od.datasets.synthetic import make_batch
    from future_od.graph import GraphedStep
    from future_od.native import functional as Fn
    q = Fn.WGRADS
    assert q.enabled
    monkeypatch.setattr(q, "_spares", [])
    monkeypatch.setattr(q, "_top_up", lambda dev: None)
    baked = len(q._baked)
    data = make_batch(2, 3, 96, 128, seed=11, max_boxes=9, device=DEV)
    data2 = make_batch(2, 3, 96, 128, seed=12, max_boxes=20, device=DEV)
    m_e, o_e = _build("bf16")
    m_g, o_g = _build("bf16")
    step = GraphedStep(m_g, o_g, warmup=2)
    seq = [data, data, data, data, data2, data]
    le = [float(_eager_step(m_e, o_e, d)) for d in seq]
    lg = [None, None, None] + [float(step(d)[1].detach()) for d in seq[3:]]
    assert step.replays == 3 and len(q._baked) == baked          # no table was baked into the graph
    for i in range(3, len(seq)):
        assert abs(lg[i] - le[i]) <= 2e-3 * max(abs(le[i]), 1.0), (i, lg[i], le[i])


def test_graph_outputs_are_the_eager_outputs_at_equal_parameters():
    """One replay against one eager forward/backward from the SAME parameters: the forward has no atomics, so the
    loss, the detections and the AP bookkeeping must be bit-identical; gradients agree to accumulation-order noise."""
    from future_od.datasets.synthetic import make_batch
    from future_od.graph import GraphedStep
    data = make_batch(2, 3, 96, 128, seed=21, max_boxes=12, device=DEV)
    m_g, o_g = _build("fp32", seed=4)
    step = GraphedStep(m_g, o_g, warmup=2)
    step(data)                                              # capture + first replay
    snap = {k: v.detach().clone() for k, v in m_g.state_dict().items()}
    post, loss, stats, od = step(data)                      # replay from `snap`
    got = (loss.detach().clone(), post["class_scores"].clone(), post["boxes"].clone(), [t.clone() for t in od])
    grads = {n: p.grad.detach().clone() for n, p in m_g.named_parameters() if p.grad is not None}
    m_e, o_e = _build("fp32", seed=4)
    m_e.load_state_dict(snap)
    o_e.zero_grad()
    post_e, _, loss_e, _, od_e = m_e(data=data, distributed=False)
    loss_e.backward()
    assert torch.equal(got[0], loss_e.detach())
    assert torch.equal(got[1], post_e["class_scores"]) and torch.equal(got[2], post_e["boxes"])
    for a, b in zip(got[3], od_e):
        assert torch.equal(a, b)
    checked = 0
    for n, p in m_e.named_parameters():
        if p.grad is None:
            continue
        g = grads[n]
        denom = float(p.grad.norm()) + 1e-12
        assert float((g - p.grad).norm()) <= 1e-4 * denom + 1e-9, n
        checked += 1
    assert checked > 50


def test_prepared_operands_follow_the_optimizer():
    """The fused AdamW kernel writes parameters through raw pointers (their `_version` does not move): the compute-
    dtype / transposed copies the GEMM kernels read must still be refreshed before the next forward.  (Round 1 shipped
    without this: every GEMM weight stayed at its initial value while only biases / norms trained.)"""
    from future_od.datasets.synthetic import make_batch
    from future_od.native import functional as Fn
    data = make_batch(2, 3, 96, 128, seed=11, max_boxes=9, device=DEV)
    m, o = _build("bf16")
    names = ["_model.detector.class_embed.weight", "_model.separate_encoder.transformer.layers.0.self_attn.mlp.0.weight",
             "_model.detector.decoder.layers.1.feedforward.3.weight"]
    params = dict(m.named_parameters())
    start = {n: params[n].detach().clone() for n in names}
    for _ in range(3):
        _eager_step(m, o, data)
    with torch.no_grad():
        m(data=data, distributed=False)                       # the forward that must see the updated weights
    for n in names:
        p = params[n]
        assert float((p.detach() - start[n]).abs().max()) > 1e-4, n           # it did train
        w = Fn.prep_linear(p, torch.bfloat16, False)
        assert torch.equal(w[:p.shape[0]], p.detach().to(torch.bfloat16)), n
        wt = Fn.prep_linear(p, torch.bfloat16, True)
        assert torch.equal(wt[:, :p.shape[0]], p.detach().to(torch.bfloat16).t()), n
    # and a conv weight of the trainable part of the backbone, with its frozen-BN scale folded in
    blk = m._model.separate_encoder.backbone.body.layer4[0]
    scale, _ = blk.bn2.scale_shift()
    w = Fn.prep_conv(blk.conv2.weight, torch.bfloat16, scale, False)
    want = (blk.conv2.weight.detach() * scale.view(-1, 1, 1, 1)).permute(0, 2, 3, 1).reshape(w.shape).to(torch.bfloat16)
    assert torch.equal(w, want)
    wt = Fn.prep_conv(blk.conv2.weight, torch.bfloat16, scale, True)                   # [Cin][taps][Cout]
    want_t = (blk.conv2.weight.detach() * scale.view(-1, 1, 1, 1)).permute(1, 2, 3, 0).reshape(wt.shape).to(torch.bfloat16)
    assert torch.equal(wt, want_t)


def test_captured_train_mode_step_draws_new_dropout_masks_every_replay():
    """Train mode with the reference's dropout 0.1: the captured step carries every dropout call's seed as a constant
    and the kernels mix in a device scalar the graph advances per replay.  The same batch replayed with lr = 0 must
    give DIFFERENT losses from replay to replay (new masks), all close to the eval-mode loss, and an eager train-mode
    forward must see masks too; backward consistency = the loss decreases under training (masks agree between the
    forward and the backward of a step, otherwise the gradients would be noise)."""
    from future_od.datasets.synthetic import make_batch
    from future_od.graph import GraphedStep
    from future_od.native import ops
    data = make_batch(2, 3, 96, 128, seed=21, max_boxes=9, device=DEV)
    model, opt = _build("bf16")
    with torch.no_grad():
        _, _, l_eval, _, _ = model(data=data, distributed=False)
    model.train()
    for g in opt.param_groups:
        g["lr"] = 0.0
    step = GraphedStep(model, opt, warmup=2)
    losses = [float(step(data)[1].detach()) for _ in range(5)]
    assert ops.DROP_BASE is not None and int(ops.DROP_BASE.item()) >= 5
    assert len({round(l, 4) for l in losses}) >= 4, losses               # new masks every replay
    for l in losses:
        assert abs(l - float(l_eval)) < 0.25 * abs(float(l_eval)), (l, float(l_eval))
    # now train: the loss on this one batch must go down over a few dozen captured steps
    for g in opt.param_groups:
        g["lr"] = 1e-4
    first = sum(float(step(data)[1].detach()) for _ in range(4)) / 4
    for _ in range(40):
        step(data)
    last = sum(float(step(data)[1].detach()) for _ in range(4)) / 4
    assert last < first, (first, last)
    ops.DROP_BASE = None


def test_captured_evaluation_pass_tracks_parameter_updates():
    """GraphedForward: the captured no-grad forward gives the eager forward's loss / detections bit for bit, and keeps
    doing so after the parameters were updated in place (its first node refreshes the prepared weight copies)."""
    from future_od.datasets.synthetic import make_batch
    from future_od.graph import GraphedForward
    data = make_batch(2, 3, 96, 128, seed=31, max_boxes=9, device=DEV)
    model, opt = _build("bf16")
    fwd = GraphedForward(model)

    def eager():
        with torch.no_grad():
            post, _, loss, stats, od = model(data=data, distributed=False)
        return float(loss), post["boxes"].clone(), [t.clone() for t in od]

    for round_ in range(2):
        post, loss, stats, od = fwd(data)
        l_e, boxes_e, od_e = eager()
        assert float(loss) == l_e, (round_, float(loss), l_e)
        assert torch.equal(post["boxes"], boxes_e)
        for a, b in zip(od, od_e):
            assert torch.equal(a, b)
        _eager_step(model, opt, data)          # parameters move (fused AdamW writes them through raw pointers)
        _eager_step(model, opt, data)
    assert fwd.replays == 2 and len(fwd._graphs) == 1


def test_optimizer_state_reload_forces_a_new_capture():
    """optimizer.load_state_dict() replaces the moment tensors with new objects; a captured step that baked the old
    addresses in must not be replayed: the next call captures anew (device step count re-synchronised) and keeps
    tracking the eagerly stepped twin."""
    from future_od.datasets.synthetic import make_batch
    from future_od.graph import GraphedStep
    data = make_batch(2, 3, 96, 128, seed=41, max_boxes=9, device=DEV)
    m_e, o_e = _build("fp32")
    m_g, o_g = _build("fp32")
    step = GraphedStep(m_g, o_g, warmup=2)
    le = [float(_eager_step(m_e, o_e, data)) for _ in range(9)]
    l4 = float(step(data)[1].detach())                       # 3 eager warm-up steps + 1 replay = optimizer steps 1..4
    assert o_g._step_no == 4 and step.replays == 1
    o_g.load_state_dict(o_g.state_dict())                    # same numbers, NEW tensors
    l7 = float(step(data)[1].detach())                       # captured anew: 2 eager warm-up steps + 1 replay = steps 5..7
    l8 = float(step(data)[1].detach())
    assert o_g._step_no == 8 and step.replays == 3 and len(step._graphs) == 1
    for got, want in ((l4, le[3]), (l7, le[6]), (l8, le[7])):
        assert abs(got - want) <= 2e-5 * max(abs(want), 1.0), (got, want)


def test_learning_rate_changes_reach_the_plan_each_graph_was_captured_with():
    """a captured step reads the lr / weight-decay table of the optimizer plan that was current at ITS capture.
    Capture signature A, change the lr, capture signature B (its warm-up rebuilds the plan), change the lr again, replay
    A: the update must use the new lr -- against the same sequence launched eagerly.  A stale table would move every
    parameter by the lr difference (8e-5) in that one step."""
    from future_od.datasets.synthetic import make_batch
    from future_od.graph import GraphedStep
    da = make_batch(2, 3, 96, 128, seed=21, max_boxes=9, device=DEV)
    db = make_batch(1, 3, 96, 128, seed=22, max_boxes=9, device=DEV)           # another batch signature: a second graph
    m_e, o_e = _build("fp32")
    m_g, o_g = _build("fp32")
    step = GraphedStep(m_g, o_g, warmup=2, rollback_warmup=True)               # one update per call, as the Trainer uses it
    plan = [(da, 1e-4), (db, 5e-5), (da, 2e-5), (db, 1e-5), (da, 1e-4)]
    for data, lr in plan:
        for opt in (o_e, o_g):
            for grp in opt.param_groups:
                grp["lr"] = lr
        _eager_step(m_e, o_e, data)
        step(data)
    assert len(step._graphs) == 2 and step.replays == len(plan)
    torch.cuda.synchronize()
    worst = max(float((pe.detach() - pg.detach()).abs().max()) for pe, pg in zip(m_e.parameters(), m_g.parameters()))
    assert worst < 3e-5, worst
    # every plan a graph was captured with is still alive (no [-8:] truncation) and is the one the graph records
    for g in step._graphs.values():
        assert g["plan"] is not None and g["plan"]["lrs"][0][0] == 1e-4


def test_failed_capture_rolls_the_warmup_steps_back_and_raises_capture_error(monkeypatch):
    """the capture's eager warm-up steps are real optimizer steps; a capture that fails AFTER them must leave
    parameters, moments and the step count as they were (the Trainer then runs the eager step on the same batch), and
    the failure must come out as CaptureError -- the only thing the Trainer's fallback catches."""
    from future_od.datasets.synthetic import make_batch
    from future_od.graph import CaptureError, GraphedStep
    data = make_batch(2, 3, 96, 128, seed=23, max_boxes=9, device=DEV)
    model, opt = _build("fp32")
    _eager_step(model, opt, data)                                               # moments exist, step count 1
    before = [p.detach().clone() for p in model.parameters()]
    moments = [opt.state[p]["exp_avg"].clone() for p in model.parameters() if p in opt.state and "exp_avg" in opt.state[p]]
    step = GraphedStep(model, opt, warmup=2, rollback_warmup=True)

    class Boom(RuntimeError):
        pass

    def broken_graph(*a, **k):
        raise Boom("no capture today")

    monkeypatch.setattr(torch.cuda, "graph", broken_graph)
    with pytest.raises(CaptureError, match="Boom"):
        step(data)
    monkeypatch.undo()
    torch.cuda.synchronize()
    assert opt._step_no == 1
    for p, b in zip(model.parameters(), before):
        assert torch.equal(p.detach(), b)
    now = [opt.state[p]["exp_avg"] for p in model.parameters() if p in opt.state and "exp_avg" in opt.state[p]]
    assert all(torch.equal(a, b) for a, b in zip(now, moments))
    # and the object still works: the next call captures and steps once
    step(data)
    assert opt._step_no == 2 and step.replays == 1


def test_matcher_failure_inside_a_replay_is_raised_at_the_next_call():
    """the device solver's failure word is only read by Python; a replayed graph runs none.  A batch whose
    annotated boxes are NaN makes the cost matrices non-finite inside the replay: the NEXT call must raise (one step late, as the
    eager path does), not the end of the epoch."""
    from future_od.datasets.synthetic import make_batch
    from future_od.graph import GraphedForward
    from future_od.native.lib import FodError
    data = make_batch(2, 3, 96, 128, seed=24, max_boxes=9, device=DEV)
    model, _ = _build("fp32")
    fwd = GraphedForward(model)
    fwd(data)
    bad = {k: v for k, v in data.items() if k != "_host_annotations"}
    bad["boxes"] = torch.full_like(data["boxes"], float("nan"))       # NaN annotations -> NaN L1 / GIoU costs
    fwd(bad)                                            # (NaN frames would not do: the first ReLU turns NaN into 0)
    torch.cuda.synchronize()
    with pytest.raises(FodError, match="non-finite"):
        fwd(data)
    fwd(data)                                           # the word was cleared: the pass works again
