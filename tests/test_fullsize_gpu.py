"""The headline size (T=6, 900x1600, ResNet-50, 6+6 layers, 128 queries) through size-independent properties: the CPU
oracle needs ~20 s per frame-sequence there, so parity proper is pinned at small sizes (test_model_gpu.py) and the
full size is held to what must be true at any size."""
import numpy as np
import pytest
import torch
from types import SimpleNamespace

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
# bf16 mode, same sample in a different batch / with the dead frames computed: tile shapes and split choices change with
# the row count, so activations round differently.  Measured on MI355X (round 3, all four workloads): logits <= 2.1e-3
# of max|logit|, boxes identical.  Bounds = two bf16 ulps of the logit range (2 x 2^-8) and one bf16 ulp of a box
# coordinate near 1 (2^-8); the round-2 bounds were 5e-2 / 2e-2.
LOGIT_BOUND, BOX_BOUND = 8e-3, 4e-3
T, H, W = 6, 900, 1600


def _build(num_images, dtype="bf16", seed=0):
    from future_od.models.st_detr import SpatioTemporalDETRArgs
    from runs._model import build_model
    torch.manual_seed(seed)
    args = SimpleNamespace(device=DEV, distributed=False, compute_dtype=dtype, num_images=num_images)
    detr = SpatioTemporalDETRArgs(num_classes=8, num_queries=128, lr_backbone=1e-4, pretrained_backbone=False)
    return build_model(args, detr).eval()


def _core(model, data):
    imu = torch.cat([data[k] for k in model._imu_keys], dim=2)
    with torch.no_grad():
        out, _ = model._model(data["video"], imu=imu)
    return out["pred_logits"].float(), out["pred_boxes"].float()


def test_full_size_samples_are_independent_and_runs_repeat():
    """No operation mixes frame-sequences (FrozenBN, LayerNorm, per-sample attention and matching): a sample's
    detections must not depend on what else is in the batch; and the same input gives the same output twice."""
    from future_od.datasets.synthetic import make_batch
    model = _build(num_images=5)
    data = make_batch(2, T, H, W, seed=11, device=DEV)
    l2, b2 = _core(model, data)
    l2b, b2b = _core(model, data)
    assert torch.equal(l2, l2b) and torch.equal(b2, b2b)                   # idempotent (no atomics in the forward pass)
    for i in range(2):
        one = {k: (v[i:i + 1] if isinstance(v, torch.Tensor) else v) for k, v in data.items() if k != "_host_annotations"}
        l1, b1 = _core(model, one)
        # tile shapes / split choices may differ with the row count, so equality is up to bf16 rounding of the activations
        dl, db = float((l1[0] - l2[i]).abs().max()) / float(l2[i].abs().max()), float((b1[0] - b2[i]).abs().max())
        print(f"batch independence, sample {i}: logits {dl:.3e} of max|logit|, boxes {db:.3e}")
        assert dl <= LOGIT_BOUND and db <= BOX_BOUND, (i, dl, db)
    assert torch.isfinite(l2).all() and (b2 >= 0).all() and (b2 <= 1).all()


def test_full_size_dead_frame_skipping_and_matching_properties():
    """With the shipped num_images = 2 only the last two past frames can reach the output: skipping the other three
    must not change it.  The matcher's assignment on the full-size outputs must be the optimal one (scipy on the same
    cost matrix), a bijection onto the targets, and every live parameter must receive a finite gradient."""
    from scipy.optimize import linear_sum_assignment
    from future_od.datasets.synthetic import make_batch
    from future_od.models.set_criterion import pack_targets
    from future_od.models.st_detr import to_detr_targets
    from future_od.native import ops
    model = _build(num_images=2)
    data = make_batch(2, T, H, W, seed=12, device=DEV)
    model._model.skip_dead_frames = True
    la, ba = _core(model, data)
    model._model.skip_dead_frames = False
    lb, bb = _core(model, data)
    model._model.skip_dead_frames = True
    dl, db = float((la - lb).abs().max()) / float(lb.abs().max()), float((ba - bb).abs().max())
    print(f"dead-frame skipping: logits {dl:.3e} of max|logit|, boxes {db:.3e}")
    assert dl <= LOGIT_BOUND and db <= BOX_BOUND, (dl, db)
    # matcher: cost on the device, assignment by the product solver vs scipy on the same matrix
    host = data["_host_annotations"]
    targets = to_detr_targets(H, W, host["active"], host["boxes"], host["classes"])
    packed = pack_targets(targets, torch.device(DEV))
    sizes = packed["sizes"]
    ld = max(max(sizes), 1)
    cost = ops.match_cost(la[None].contiguous(), ba[None].contiguous(), packed["labels"], packed["boxes"],
                          packed["offset"], ld, 2.0, 5.0, 2.0).cpu()
    ours = ops.lap_solve_batch_host(cost.view(2, 128, ld), sizes)
    for b in range(2):
        r, c = linear_sum_assignment(cost[0, b, :, :sizes[b]].numpy())
        want = np.full(128, -1, dtype=np.int32)
        want[r] = c
        assert np.array_equal(ours[b].numpy(), want), b                    # bit-exact indices
        assert sorted(ours[b][ours[b] >= 0].tolist()) == list(range(sizes[b]))   # every target exactly once
    # one full step: finite loss, finite gradients everywhere they are expected, none on the frozen stem / layer1
    model.zero_grad(set_to_none=True)
    out, _, loss, stats, od = model(data=data, distributed=False)
    loss.backward()
    assert torch.isfinite(loss)
    for n, p in model.named_parameters():
        if not p.requires_grad:
            continue
        assert p.grad is not None and torch.isfinite(p.grad).all(), n
    frozen = [n for n, p in model.named_parameters() if ".layer1." in n or n.endswith("body.conv1.weight")]
    assert frozen and all(not dict(model.named_parameters())[n].requires_grad for n in frozen)


def test_full_size_queued_weight_gradients_equal_the_direct_ones():
    """The weight gradients a captured step queues (short ones: fod_gemm_tn_multi, chained for layers used several
    times; long ones -- the encoder's and the memory side's Linear layers, 14 500 rows -- fod_gemm_tn_multi_long) against
    the same backward pass with every gradient launched where it is produced, on the full-size model: the queue's wiring
    (slices of packed in_proj gradients, grouped projections, shared layers) is exercised where the long path applies.
    Equal up to f32 summation order (the long launch cuts M differently; atomics)."""
    from future_od.datasets.synthetic import make_batch
    from future_od.native import functional as Fn
    model = _build(num_images=5)
    data = make_batch(2, T, H, W, seed=13, device=DEV)
    q = Fn.WGRADS
    was = q.enabled, q.eager
    names = [n for n, p in model.named_parameters() if p.requires_grad]

    def grads(queue):
        q.enabled, q.eager = queue, queue
        for p in model.parameters():
            p.grad = None
        _, _, loss, _, _ = model(data=data, distributed=False)
        loss.backward()
        torch.cuda.synchronize()
        return {n: p.grad.detach().float().clone() for n, p in model.named_parameters() if p.grad is not None}, float(loss.detach())

    try:
        ref, l0 = grads(False)
        l_before, c_before = q.launches, q.carried
        got, l1 = grads(True)
        launched, carried = q.launches - l_before, q.carried - c_before
    finally:
        q.enabled, q.eager = was
    assert l0 == l1
    assert launched <= 4 and carried >= 150, (launched, carried)        # ~130 short + ~40 long jobs in a few launches
    assert set(got) == set(ref) and len(got) == len(names)
    worst = ("", 0.0)
    for n in names:
        a, b = got[n], ref[n]
        assert torch.isfinite(a).all(), n
        scale = float(b.abs().max())
        err = float((a - b).abs().max())
        # f32 sums of <= 14 500 bf16 products in another order: a few ulp of the largest partial sums
        assert err <= 2e-4 * max(scale, 1e-6) + 1e-7, (n, err, scale)
        if scale > 0 and err / scale > worst[1]:
            worst = (n, err / scale)
    print("largest relative difference", worst)
