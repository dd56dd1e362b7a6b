"""HIP path against the CPU oracle AT THE HEADLINE EXTENT ("do this" 3): B=1, T=6, 900x1600, ResNet-50,
6+6 layers, K=5, 128 queries -- 29x50 maps, 1450 tokens per frame, ragged tiles in every stage
(reference future_od/models/st_detr.py:98-167, paper.py:448-485).

  * fp32 mode (exact-f32 MFMA): pred_logits, pred_boxes, every aux level and the set loss against
    oracle.core_forward / total_loss on the same weights and inputs, max|err| <= 1e-3 * max|ref| per tensor
    (BASELINE.json north_star: "within 1e-3 rel fp32"); Hungarian indices bit-exact against the oracle's matcher
    (scipy) on the PRODUCT's own outputs, and their agreement with the oracle's end-to-end indices is printed.
  * bf16 mode (the benched dtype) at the same extent: its deviation from the fp32 oracle is MEASURED, printed and
    held to the bounds DESIGN.md states (logits <= 1.5e-2 of range, boxes <= 1e-2 abs, loss <= 0.5 %; measured on the
    box: 3.6e-3, 2.2e-3, 4e-5).

The oracle needs ~10-20 s of the box's host cores for this one forward; it runs once per session (module fixture)."""
import pytest
import torch

from future_od.datasets.synthetic import make_batch
from oracle import criterion as ocrit
from oracle import stdetr as O
from oracle.stdetr import Config

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
T_, H_, W_, SEED = 6, 900, 1600, 21
CFG = Config(backbone="resnet50", enc_layers=6, dec_layers=6, num_images=5)


@pytest.fixture(scope="module")
def reference():
    """Oracle forward + set loss on the host (checker only)."""
    import os
    torch.set_num_threads(min(os.cpu_count() or 1, 64))
    data = make_batch(1, T_, H_, W_, seed=SEED, max_boxes=40)
    sd = O.make_state_dict(CFG, SEED)
    with torch.no_grad():
        ref = O.core_forward(sd, CFG, data["video"], O.imu_from_data(data), skip_dead=True)
        loss, _, _ = ocrit.total_loss(CFG, ref, data)
    targets = ocrit.to_detr_targets(H_, W_, data["active"], data["boxes"], data["classes"])
    idx = ocrit.hungarian_match(CFG, ref["pred_logits"], ref["pred_boxes"], targets)
    return data, ref, float(loss), targets, idx


def _product(dtype, data_cpu):
    from test_model_gpu import build_product
    model, _ = build_product(CFG, dtype, SEED)
    data = {k: (v.to(DEV) if isinstance(v, torch.Tensor) else v) for k, v in data_cpu.items()}
    with torch.no_grad():
        out, _, loss, stats, od = model(data=data, distributed=False)
        raw, _ = model._model(data["video"], imu=torch.cat([data[k] for k in model._imu_keys], dim=2))
    torch.cuda.synchronize()
    return model, raw, float(loss)


def _err(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).abs().max()), float((a - b).abs().mean()), float(b.abs().max())


def test_fp32_mode_matches_the_oracle_at_the_headline_extent(reference):
    data_cpu, ref, ref_loss, targets, ref_idx = reference
    model, raw, loss = _product(torch.float32, data_cpu)
    levels = [("final", raw, ref)] + [(f"aux{i}", a, r) for i, (a, r) in enumerate(zip(raw["aux_outputs"], ref["aux_outputs"]))]
    for name, got, want in levels:
        for key in ("pred_logits", "pred_boxes"):
            err, mean, scale = _err(got[key], want[key])
            print(f"fp32 {name} {key}: max err {err:.3e} (mean {mean:.3e}) of max|ref| {scale:.3e} -> {err / scale:.2e}")
            assert err <= 1e-3 * scale, (name, key, err, scale)
    print(f"fp32 loss {loss:.6f} vs oracle {ref_loss:.6f}: rel {abs(loss - ref_loss) / abs(ref_loss):.2e}")
    assert abs(loss - ref_loss) <= 1e-3 * abs(ref_loss)
    # Hungarian assignment: bit-exact on the product's own outputs (device cost + device LAP vs the oracle's scipy)
    mine = model._criterion.matcher({"pred_logits": raw["pred_logits"], "pred_boxes": raw["pred_boxes"]},
                                    [{k: v.to(DEV) for k, v in t.items()} for t in targets])
    own = ocrit.hungarian_match(CFG, raw["pred_logits"].float().cpu(), raw["pred_boxes"].float().cpu(), targets)
    for (i, j), (ri, rj) in zip(mine, own):
        assert torch.equal(i.cpu(), ri) and torch.equal(j.cpu(), rj)
    same = sum(int(torch.equal(i.cpu(), ri) and torch.equal(j.cpu(), rj)) for (i, j), (ri, rj) in zip(mine, ref_idx))
    print(f"end-to-end assignment equal to the oracle's for {same}/{len(ref_idx)} samples")
    assert same == len(ref_idx)


def test_bf16_mode_deviation_at_the_headline_extent(reference):
    """The benched dtype is a precision trade: the deviation is recorded (stdout, -s) and bounded, not called parity."""
    data_cpu, ref, ref_loss, _, _ = reference
    _, raw, loss = _product(torch.bfloat16, data_cpu)
    e_l, m_l, s_l = _err(raw["pred_logits"], ref["pred_logits"])
    e_b, m_b, _ = _err(raw["pred_boxes"], ref["pred_boxes"])
    rel_loss = abs(loss - ref_loss) / abs(ref_loss)
    print(f"bf16 vs fp32 oracle at 900x1600: logits max {e_l:.3e} mean {m_l:.3e} (range {s_l:.3e}: {e_l / s_l:.2e}), "
          f"boxes max {e_b:.3e} mean {m_b:.3e}, loss {loss:.5f} vs {ref_loss:.5f} ({rel_loss:.2e})")
    assert e_l <= 1.5e-2 * s_l
    assert e_b <= 1e-2
    assert rel_loss <= 5e-3
