"""Positional tables, reference-point / box-head kernels, matcher cost, LAP, set losses and the AP
bookkeeping kernel against the CPU oracle (oracle/) and the committed golden fixtures."""
import numpy as np
import pytest
import torch

from oracle import criterion as ocrit
from oracle import stdetr as O
from oracle import thirdparty as tp

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from future_od.native import ops

DEV = "cuda:0"


def close(a, b, atol=1e-5, rtol=1e-5):
    a = a.detach().float().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = b.detach().float().cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b)
    np.testing.assert_allclose(a, b, atol=atol, rtol=rtol)


@pytest.mark.parametrize("hw", [(7, 7), (25, 42), (29, 50)])
def test_posenc_table(golden, hw):
    h, w = hw
    t = ops.posenc_table(h, w, 256, torch.float32, DEV)            # [h*w, C]
    ref = O.spatial_pos_table(h, w, 256).flatten(1).t()
    close(t, ref, atol=2e-6, rtol=0)
    g = golden("g1_posenc")
    flat = t.t().reshape(256, h, w).reshape(-1).cpu()               # golden is (C,h,w) order
    close(flat[torch.from_numpy(g[f"idx_{h}x{w}"])], g[f"val_{h}x{w}"], atol=2e-6, rtol=0)


def test_posenc_temporal(golden):
    g = golden("g1_posenc")
    offs = torch.from_numpy(g["st_offsets"])
    sp = O.spatial_pos_table(5, 6, 64)[None, None]
    t = ops.posenc_temporal(2, 3, 64, torch.float32, DEV, offsets=offs.to(DEV))
    close(sp + t.cpu()[:, :, :, None, None], g["st_full"], atol=2e-6, rtol=0)
    t = ops.posenc_temporal(2, 3, 64, torch.float32, DEV)
    close(sp + t.cpu()[:, :, :, None, None], g["st_noffs"], atol=2e-6, rtol=0)


def test_refpoint_sine_and_box_finish():
    g = torch.Generator().manual_seed(3)
    R, D, Lv = 40, 256, 3
    logit = torch.randn(R, 2, generator=g).requires_grad_(True)
    t = (torch.randn(Lv, R, 4, generator=g)).requires_grad_(True)
    ref = logit.sigmoid()
    sine = O.query_sine_embed(ref[:, None, :], D)[:, 0]
    rl = tp.inverse_sigmoid(ref)
    boxes = torch.cat([t[..., :2] + rl, t[..., 2:]], -1).sigmoid()
    dsine, dboxes = torch.randn(R, D, generator=g), torch.randn(Lv, R, 4, generator=g)
    ((sine * dsine).sum() + (boxes * dboxes).sum()).backward()
    ref_d, sine_d = ops.refpoint_sine_fwd(logit.detach().to(DEV), D)
    close(ref_d, ref); close(sine_d, sine, atol=2e-5)
    boxes_d = ops.box_finish_fwd(t.detach().to(DEV), ref_d, Lv)
    close(boxes_d, boxes, atol=1e-6)
    dt_d, dref_d = ops.box_finish_bwd(dboxes.to(DEV), boxes_d, ref_d, torch.float32)
    close(dt_d, t.grad, atol=1e-6, rtol=1e-4)
    dlogit = ops.refpoint_sine_bwd(dsine.to(DEV), ref_d, dref_d)
    close(dlogit, logit.grad, atol=2e-4, rtol=1e-4)


def _targets(nbs, seed):
    g = torch.Generator().manual_seed(seed)
    out = []
    for nb in nbs:
        cxcy = torch.rand(nb, 2, generator=g) * 0.6 + 0.2
        wh = torch.rand(nb, 2, generator=g) * 0.3 + 0.02
        out.append({"labels": torch.randint(0, 8, (nb,), generator=g), "boxes": torch.cat([cxcy, wh], 1)})
    return out


@pytest.mark.parametrize("case", [(2, 128, [7, 23]), (3, 16, [0, 1, 40]), (1, 8, [0]), (2, 32, [32, 5])])
def test_matcher_and_losses(case):
    B, M, nbs = case
    cfg = O.Config(dec_layers=3)
    Lv = 3
    g = torch.Generator().manual_seed(5)
    logits = (torch.randn(Lv, B, M, 8, generator=g) * 2 - 2).requires_grad_(True)
    boxes = (torch.rand(Lv, B, M, 4, generator=g) * 0.5 + 0.1).requires_grad_(True)
    targets = _targets(nbs, 6)
    tl = torch.cat([t["labels"] for t in targets])
    tb = torch.cat([t["boxes"] for t in targets])
    toff = torch.tensor([0] + list(np.cumsum(nbs)), dtype=torch.int32)
    ld = max(max(nbs), 1)
    cost = ops.match_cost(logits.detach().to(DEV), boxes.detach().to(DEV), tl.to(DEV), tb.to(DEV), toff.to(DEV),
                          ld, 2.0, 5.0, 2.0).cpu()
    matches = ops.lap_solve_batch_host(cost.view(Lv * B, M, ld), [nbs[b] for _ in range(Lv) for b in range(B)])
    matches = matches.view(Lv, B, M)
    gmatch = torch.full((Lv, B, M), -1, dtype=torch.int32)
    for lv in range(Lv):
        idx, Cref = ocrit.hungarian_match(cfg, logits[lv].detach(), boxes[lv].detach(), targets, return_cost=True)
        for b, (i, j) in enumerate(idx):
            if nbs[b]:
                close(cost[lv, b, :, :nbs[b]], Cref.split(nbs, -1)[b][b], atol=2e-5, rtol=1e-5)
            mine = matches[lv, b]
            qi = torch.nonzero(mine >= 0).flatten()
            assert torch.equal(qi, i), (lv, b)                          # bit-exact assignment, rows ascending
            assert torch.equal(mine[qi].long(), j), (lv, b)
            gmatch[lv, b, i] = (j + int(toff[b])).int()
    nb_total = max(float(sum(nbs)), 1.0)
    out = ops.set_loss_fwd(logits.detach().to(DEV), boxes.detach().to(DEV), gmatch.to(DEV), tl.to(DEV), tb.to(DEV),
                           toff.to(DEV), nb_total, 0.25).cpu()
    total = 0
    w = torch.rand(Lv, 3, generator=g) + 0.5
    for lv in range(Lv):
        outputs = {"pred_logits": logits[lv], "pred_boxes": boxes[lv]}
        ld_ = ocrit.set_criterion(cfg, outputs, targets)
        close(out[lv, 0], ld_["loss_ce"], atol=1e-5, rtol=2e-5)
        close(out[lv, 1], ld_["loss_bbox"], atol=1e-5, rtol=2e-5)
        close(out[lv, 2], ld_["loss_giou"], atol=1e-5, rtol=2e-5)
        close(out[lv, 3], ld_["cardinality_error"])
        close(out[lv, 4], ld_["class_error"], atol=1e-4)
        total = total + w[lv, 0] * ld_["loss_ce"] + w[lv, 1] * ld_["loss_bbox"] + w[lv, 2] * ld_["loss_giou"]
    total.backward()
    dl, db = ops.set_loss_bwd(logits.detach().to(DEV), boxes.detach().to(DEV), gmatch.to(DEV), tl.to(DEV),
                              tb.to(DEV), w.to(DEV), nb_total, 0.25)
    close(dl, logits.grad, atol=1e-6, rtol=1e-4)
    close(db, boxes.grad, atol=1e-5, rtol=1e-4)


def test_post_proc_and_od_map(golden):
    g = golden("g9_odmap")
    for ci in range(3):
        scores = torch.from_numpy(g[f"c{ci}_scores"]).to(DEV)
        out = ops.od_map(scores, torch.from_numpy(g[f"c{ci}_pboxes"]).to(DEV),
                         torch.from_numpy(g[f"c{ci}_aboxes"]).to(DEV), torch.from_numpy(g[f"c{ci}_aclasses"]).to(DEV),
                         torch.from_numpy(g[f"c{ci}_active"]).to(DEV), (448, 800))
        np.testing.assert_array_equal(out[0].cpu().numpy(), g[f"c{ci}_confs"])
        np.testing.assert_array_equal(out[1].cpu().numpy(), g[f"c{ci}_is_positive"])
        np.testing.assert_array_equal(out[2].cpu().numpy(), g[f"c{ci}_size_categories"])
        np.testing.assert_array_equal(out[3].cpu().numpy(), g[f"c{ci}_num_annos"])
    gg = torch.Generator().manual_seed(9)
    logits, boxes = torch.randn(2, 1, 16, 8, generator=gg), torch.rand(2, 1, 16, 4, generator=gg)
    s, bp = ops.post_proc(logits.to(DEV), boxes.to(DEV), 448, 800)
    sr = logits.sigmoid()
    close(s, torch.cat([sr, sr.max(3, keepdim=True)[0]], 3), atol=1e-6)
    bb = boxes * torch.tensor([800.0, 448.0, 800.0, 448.0])
    close(bp, torch.cat([bb[..., :2] - 0.5 * bb[..., 2:], bb[..., :2] + 0.5 * bb[..., 2:]], -1), atol=1e-4)


def test_od_map_with_nan_scores_stays_in_range():
    """Regression (round 2, DESIGN.md 3: "a NaN score must not become a wild index"): fod_od_map ranks by comparisons;
    NaN scores all claimed rank 0 and left the other slots uninitialised -- a GPU memory fault one kernel after a
    diverged forward.  NaN ranks last: the launch completes, confidences of real detections stay finite and ordered,
    every output has its shape, nothing faults."""
    gg = torch.Generator().manual_seed(3)
    B, M, C1, N = 2, 128, 9, 12
    scores = torch.rand(B, M, C1, generator=gg)
    scores[0, ::3] = float("nan")                       # a third of one sample's detections ...
    scores[1] = float("nan")                            # ... and all of the other's
    boxes = torch.rand(B, M, 4, generator=gg) * 100
    boxes[..., 2:] += boxes[..., :2] + 4
    ab = torch.rand(B, N, 4, generator=gg) * 100
    ab[..., 2:] += ab[..., :2] + 4
    out = ops.od_map(scores.to(DEV), boxes.to(DEV), ab.to(DEV), torch.randint(0, 8, (B, N), generator=gg).to(DEV),
                     torch.ones(B, N, dtype=torch.int64).to(DEV), (448, 800))
    torch.cuda.synchronize()
    confs = out[0].cpu()                                # [T, C1, B * 50]: sample 0 first
    assert confs.shape == (10, C1, B * 50) and out[1].shape == confs.shape and out[3].shape == (C1, 4)
    first = confs[0, :, :50]                            # sample 0: 85 finite scores per class -> its top 50 are all finite
    assert torch.isfinite(first).all() and (first[:, :-1] >= first[:, 1:]).all()
    assert int(out[3].sum()) >= 0


def test_group_linear_function_matches_separate_linears():
    """GroupLinearFn (one launch for P Linear layers of one input, bf16) against P LinearFn calls: same outputs
    and gradients up to bf16 rounding of the differently ordered sums; unused outputs get zero gradient."""
    import torch.nn as nn
    from future_od.native import functional as Fn
    torch.manual_seed(0)
    P, D, K, rows = 4, 256, 256, 200
    lins = [nn.Linear(K, D).to("cuda:0") for _ in range(P)]
    x0 = torch.randn(2, rows // 2, K, device="cuda:0").to(torch.bfloat16)
    outs = {}
    for mode in ("group", "separate"):
        for m in lins:
            m.zero_grad()
        x = x0.clone().requires_grad_(True)
        ys = Fn.group_linear(x, lins) if mode == "group" else [Fn.linear(x, m.weight, m.bias) for m in lins]
        assert all(y.shape == (2, rows // 2, D) for y in ys)
        loss = sum((i + 1.0) * (y.float() ** 2).mean() for i, y in enumerate(ys[:-1]))      # last output unused
        loss.backward()
        outs[mode] = ([y.detach().float() for y in ys], x.grad.float(),
                      [m.weight.grad.clone() if m.weight.grad is not None else None for m in lins],
                      [m.bias.grad.clone() if m.bias.grad is not None else None for m in lins])
    a, b = outs["group"], outs["separate"]
    for ya, yb in zip(a[0], b[0]):
        assert torch.allclose(ya, yb, rtol=2e-2, atol=2e-2)
    assert torch.allclose(a[1], b[1], rtol=3e-2, atol=1e-4)
    for i in range(P - 1):
        assert torch.allclose(a[2][i], b[2][i], rtol=3e-2, atol=1e-5), i
        assert torch.allclose(a[3][i], b[3][i], rtol=3e-2, atol=1e-5), i
    assert float(a[2][P - 1].abs().max()) == 0.0 and b[2][P - 1] is None


def test_dropout_is_stateless_and_self_consistent():
    """fod_dropout keeps each element with probability 1 - p, scales by 1 / (1 - p), depends only on (seed, index),
    and DropoutFn's backward applies the forward's mask to the gradient."""
    from future_od.native import functional as Fn
    for dtype in (torch.float32, torch.bfloat16):
        x = torch.ones(512, 1024, device=DEV, dtype=dtype)
        y1 = ops.dropout(x, 0.1, 1234)
        y2 = ops.dropout(x, 0.1, 1234)
        y3 = ops.dropout(x, 0.1, 1235)
        assert torch.equal(y1, y2) and not torch.equal(y1, y3)
        kept = (y1 != 0).float().mean().item()
        assert abs(kept - 0.9) < 3e-3, kept
        vals = y1[y1 != 0].float()
        assert torch.allclose(vals, torch.full_like(vals, 1 / 0.9), rtol=1e-2)
        # rows / columns are not systematically favoured
        assert (y1 != 0).float().mean(0).std().item() < 0.03 and (y1 != 0).float().mean(1).std().item() < 0.03
    x = torch.randn(64, 256, device=DEV, requires_grad=True)
    seed = Fn.DROP_SEEDS.next()
    y = Fn.DropoutFn.apply(x, 0.25, seed)
    y.backward(torch.ones_like(y))
    mask = (y.detach() != 0).float() / 0.75
    assert torch.allclose(x.grad, mask, rtol=1e-6, atol=0) and torch.allclose(y.detach(), x.detach() * mask, rtol=1e-6)
    assert Fn.dropout(x, 0.25, training=False) is x and Fn.dropout(x, 0.0, training=True) is x


def _criterion_inputs(seed=0, B=3, M=32, Lv=3, C=8):
    g = torch.Generator().manual_seed(seed)
    logits = (torch.randn(Lv, B, M, C, generator=g) * 2 - 2).to(DEV).requires_grad_(True)
    boxes = (torch.rand(Lv, B, M, 4, generator=g) * 0.5 + 0.2).to(DEV).requires_grad_(True)
    targets = []
    for b, nb in enumerate([5, 0, 17][:B]):
        xy = torch.rand(nb, 2, generator=g) * 0.5 + 0.2
        targets.append({"labels": torch.randint(0, C, (nb,), generator=g),
                        "boxes": torch.cat([xy, torch.rand(nb, 2, generator=g) * 0.2 + 0.05], 1)})
    return logits, boxes, targets


def test_async_matcher_equals_host_synchronous_matcher(monkeypatch):
    """The stream-parking matcher (fod_stream_wait_flag + fod_match_after_event on a worker thread) must produce
    exactly the matches, losses and gradients of the path that blocks the host on the cost matrix."""
    from future_od.models.set_criterion import SetCriterion, build_matcher, _async_lap
    from future_od.models.st_detr import SpatioTemporalDETRArgs
    crit = SetCriterion(8, build_matcher(SpatioTemporalDETRArgs(num_classes=8)), {}, 0.25,
                        ["labels", "boxes", "cardinality"], "per level")
    res = {}
    for mode in ("0", "1", "2"):                              # host sync / adaptive / always through the worker
        monkeypatch.setenv("FOD_ASYNC_MATCH", mode)
        tables = []
        for it in range(3):                                   # several tickets in flight one after the other
            logits, boxes, targets = _criterion_inputs(seed=it)
            out = crit({"_stacked": (logits, boxes)}, targets, distributed=False)
            out.table[:, :3].sum().backward()
            tables.append((out.table.detach().cpu(), logits.grad.cpu(), boxes.grad.cpu()))
        res[mode] = tables
    _async_lap(torch.device(DEV)).join()
    for mode in ("1", "2"):
        for (t0, gl0, gb0), (t1, gl1, gb1) in zip(res["0"], res[mode]):
            assert torch.equal(t0, t1) and torch.equal(gl0, gl1) and torch.equal(gb0, gb1), mode


def test_async_matcher_failure_releases_the_stream_and_raises_late(monkeypatch):
    """A cost matrix the assignment solver rejects (non-finite) must not leave the stream parked: the worker writes
    'no match' and sets the flag; the error surfaces at the next join."""
    from future_od.models.set_criterion import SetCriterion, build_matcher, _async_lap
    from future_od.models.st_detr import SpatioTemporalDETRArgs
    from future_od.native.lib import FodError
    monkeypatch.setenv("FOD_ASYNC_MATCH", "2")                 # through the worker thread
    crit = SetCriterion(8, build_matcher(SpatioTemporalDETRArgs(num_classes=8)), {}, 0.25,
                        ["labels", "boxes", "cardinality"], "per level")
    logits, boxes, targets = _criterion_inputs(seed=5)
    with torch.no_grad():
        bad = boxes.detach().clone()
        bad[0, 0, 0, 0] = float("nan")
    out = crit({"_stacked": (logits.detach(), bad)}, targets, distributed=False)
    torch.cuda.synchronize()                                   # returns: the stream was released
    with pytest.raises(FodError):
        _async_lap(torch.device(DEV)).join()
    logits2, boxes2, targets2 = _criterion_inputs(seed=6)      # and the next submission works
    out2 = crit({"_stacked": (logits2, boxes2)}, targets2, distributed=False)
    assert torch.isfinite(out2.table[:, :3]).all()


@pytest.mark.parametrize("case", [(128, [23, 7], 40), (128, [0, 1], 8), (128, [128, 127], 128), (128, [200, 129], 256),
                                  (16, [40, 3], 64), (37, [1, 36], 48), (128, [256, 5], 256), (8, [8, 8], 8)])
def test_device_lap_is_bit_identical_to_the_host_solver_and_scipy(case):
    """fod_lap_solve_batch_dev (one wavefront per problem, target counts read on the device) against
    fod_lap_solve_batch_host (scipy's algorithm) and scipy itself: identical assignments, exact ties included."""
    from scipy.optimize import linear_sum_assignment
    M, sizes, ld = case
    B, Lv = len(sizes), 3
    rng = np.random.default_rng(M * 7919 + sum(sizes) * 31 + ld)
    cost = torch.from_numpy(rng.standard_normal((Lv, B, M, ld)).astype(np.float32) * 3)
    cost[0, :, :, 1] = cost[0, :, :, 0]                 # a duplicated column: exact ties
    cost[1] = torch.round(cost[1])                       # small-integer costs: many ties
    cost[2, :, 1::2] = cost[2, :, 0::2][:, : cost[2, :, 1::2].shape[1]]   # duplicated rows
    off = [0]
    for s in sizes:
        off.append(off[-1] + s)
    off_dev = torch.tensor(off, dtype=torch.int32, device=DEV)
    status = torch.zeros(1, dtype=torch.int32, device=DEV)
    got = ops.lap_solve_batch_dev(cost.to(DEV), off_dev, status).cpu()
    assert int(status.item()) == 0
    host = ops.lap_solve_batch_host(cost.view(Lv * B, M, ld).contiguous(), sizes * Lv, threads=2).view(Lv, B, M)
    for lv in range(Lv):
        for b in range(B):
            want = torch.where(host[lv, b] >= 0, host[lv, b] + off[b], host[lv, b])
            assert torch.equal(got[lv, b], want), (lv, b, sizes[b])
            n = sizes[b]
            if n:
                i, j = linear_sum_assignment(cost[lv, b, :, :n].numpy())
                mine = got[lv, b]
                assert np.array_equal(np.nonzero(mine.numpy() >= 0)[0], i)
                assert np.array_equal(mine[mine >= 0].numpy() - off[b], j)


def test_device_lap_flags_non_finite_costs():
    cost = torch.zeros(1, 1, 16, 8)
    cost[0, 0, 3, 2] = float("nan")
    off = torch.tensor([0, 5], dtype=torch.int32, device=DEV)
    status = torch.zeros(1, dtype=torch.int32, device=DEV)
    got = ops.lap_solve_batch_dev(cost.to(DEV), off, status)
    assert int(status.item()) != 0 and bool((got == -1).all())


def test_device_target_path_equals_host_matcher(monkeypatch):
    """Targets packed on the device (fod_pack_targets), matched by the device solver and normalised by a device-side
    `num_boxes` must give exactly the loss table and gradients of the host path (to_detr_targets + pack_targets +
    host solver + host num_boxes)."""
    from future_od.models.set_criterion import SetCriterion, build_matcher
    from future_od.models.st_detr import SpatioTemporalDETRArgs, to_detr_targets
    monkeypatch.setenv("FOD_ASYNC_MATCH", "0")
    crit = SetCriterion(8, build_matcher(SpatioTemporalDETRArgs(num_classes=8)), {}, 0.25,
                        ["labels", "boxes", "cardinality"], "per level")
    H, W, N = 96, 160, 32
    for seed, counts in ((1, [5, 0, 17]), (2, [32, 1, 9]), (3, [0, 0, 0])):
        logits, boxes, _ = _criterion_inputs(seed=seed)
        B = logits.shape[1]
        g = torch.Generator().manual_seed(seed)
        ab = torch.zeros(B, N, 4)
        x0 = torch.rand(B, N, generator=g) * (W - 20)
        y0 = torch.rand(B, N, generator=g) * (H - 20)
        ab[..., 0], ab[..., 1] = x0, y0
        ab[..., 2] = x0 + 4 + torch.rand(B, N, generator=g) * 15
        ab[..., 3] = y0 + 4 + torch.rand(B, N, generator=g) * 15
        cls = torch.randint(0, 8, (B, N), generator=g)
        act = torch.zeros(B, N, dtype=torch.int64)
        for b in range(B):
            act[b, torch.randperm(N, generator=g)[:counts[b]]] = 1          # scattered active rows
        targets = to_detr_targets(H=H, W=W, anno_active=act, anno_boxes=ab, anno_classes=cls)
        out_h = crit({"_stacked": (logits, boxes)}, targets, distributed=False)
        out_h.table[:, :3].sum().backward()
        ref = (out_h.table.detach().cpu(), logits.grad.clone().cpu(), boxes.grad.clone().cpu())
        logits.grad = None
        boxes.grad = None
        packed = ops.pack_targets_dev(ab.to(DEV), cls.to(DEV), act.to(DEV), H, W)
        tot = sum(counts)
        assert packed["offset"].cpu().tolist() == [0] + list(np.cumsum(counts)) and float(packed["count"].item()) == tot
        if tot:
            assert torch.equal(packed["labels"][:tot].cpu(), torch.cat([t["labels"] for t in targets]))
            assert torch.equal(packed["boxes"][:tot].cpu(), torch.cat([t["boxes"] for t in targets]))
        nb = crit.device_num_boxes(packed["count"], distributed=False)
        out_d = crit({"_stacked": (logits, boxes)}, None, distributed=False, packed=packed, num_boxes=nb)
        out_d.table[:, :3].sum().backward()
        assert torch.equal(out_d.table.detach().cpu(), ref[0]), seed
        assert torch.equal(logits.grad.cpu(), ref[1]) and torch.equal(boxes.grad.cpu(), ref[2]), seed
