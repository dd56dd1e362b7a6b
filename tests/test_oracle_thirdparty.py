"""Secondary cross-check of our authoring of the absent ConditionalDETR arithmetic (oracle/thirdparty.py,
"parity unpinned") against an independent descendant of the same DETR code that ships inside the
`transformers` wheel on this image.  Used only as a sanity check on random inputs, never as "the
reference"."""
import pytest
import torch

from oracle import thirdparty as tp

lod = pytest.importorskip("transformers.loss.loss_for_object_detection")


def test_giou_and_focal_match_transformers():
    g = torch.Generator().manual_seed(0)
    a = tp.box_cxcywh_to_xyxy(torch.cat([torch.rand(40, 2, generator=g), torch.rand(40, 2, generator=g) * 0.4 + 0.01], 1))
    b = tp.box_cxcywh_to_xyxy(torch.cat([torch.rand(17, 2, generator=g), torch.rand(17, 2, generator=g) * 0.4 + 0.01], 1))
    torch.testing.assert_close(tp.generalized_box_iou(a, b), lod.generalized_box_iou(a, b))
    x = torch.randn(3, 50, 8, generator=g) * 3
    t = (torch.rand(3, 50, 8, generator=g) < 0.1).float()
    torch.testing.assert_close(tp.sigmoid_focal_loss(x, t, 7.0, alpha=0.25, gamma=2),
                               lod.sigmoid_focal_loss(x, t, 7.0, alpha=0.25, gamma=2))


def test_matcher_matches_transformers_focal_matcher():
    ldd = pytest.importorskip("transformers.loss.loss_deformable_detr")
    g = torch.Generator().manual_seed(1)
    logits = torch.randn(2, 30, 8, generator=g) * 2 - 2
    boxes = torch.cat([torch.rand(2, 30, 2, generator=g) * 0.6 + 0.2, torch.rand(2, 30, 2, generator=g) * 0.3 + 0.02], 2)
    targets = []
    for nb in (5, 11):
        targets.append({"class_labels": torch.randint(0, 8, (nb,), generator=g),
                        "boxes": torch.cat([torch.rand(nb, 2, generator=g) * 0.6 + 0.2,
                                            torch.rand(nb, 2, generator=g) * 0.3 + 0.02], 1)})
    theirs = ldd.DeformableDetrHungarianMatcher(class_cost=2.0, bbox_cost=5.0, giou_cost=2.0)(
        {"logits": logits, "pred_boxes": boxes}, targets)
    ours = tp.HungarianMatcher(2.0, 5.0, 2.0)({"pred_logits": logits, "pred_boxes": boxes},
                                             [{"labels": t["class_labels"], "boxes": t["boxes"]} for t in targets])
    for (i, j), (oi, oj) in zip(theirs, ours):
        assert torch.equal(i, oi) and torch.equal(j, oj)


def test_projection_free_mha_matches_torch_mha_core_with_identity_projections():
    """ConditionalDETR's attention.py is torch's multi_head_attention_forward with the input projections removed
    (out_proj kept, `vdim` = value width).  Second source: torch's own function with IDENTITY q/k/v projection
    weights and no input bias -- same head split, same head_dim^-0.5 query scaling, same softmax, same out_proj;
    also the head-averaged weights it returns."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(3)
    for (L, S, B, E, Ev, H) in ((7, 11, 2, 64, 64, 8), (5, 9, 3, 128, 64, 8), (4, 4, 1, 512, 256, 8)):
        q = torch.randn(L, B, E, generator=g)
        k = torch.randn(S, B, E, generator=g)
        v = torch.randn(S, B, Ev, generator=g)
        ow, ob = torch.randn(Ev, Ev, generator=g) / Ev ** 0.5, torch.randn(Ev, generator=g)
        out, w = tp.projection_free_mha(q, k, v, H, ow, ob)
        if E == Ev:
            ref, wref = F.multi_head_attention_forward(
                q, k, v, E, H, in_proj_weight=None, in_proj_bias=None, bias_k=None, bias_v=None, add_zero_attn=False,
                dropout_p=0.0, out_proj_weight=ow, out_proj_bias=ob, training=False, need_weights=True,
                use_separate_proj_weight=True, q_proj_weight=torch.eye(E), k_proj_weight=torch.eye(E),
                v_proj_weight=torch.eye(E))
            torch.testing.assert_close(out, ref, rtol=1e-5, atol=1e-5)
            torch.testing.assert_close(w, wref, rtol=1e-5, atol=1e-6)
        # vdim != embed (the conditional cross-attention: 512-wide q/k, 256-wide v): closed form per head
        hd, hv = E // H, Ev // H
        qh = (q * hd ** -0.5).reshape(L, B * H, hd).transpose(0, 1)
        kh = k.reshape(S, B * H, hd).transpose(0, 1)
        vh = v.reshape(S, B * H, hv).transpose(0, 1)
        p = torch.softmax(qh @ kh.transpose(1, 2), -1)
        manual = (p @ vh).transpose(0, 1).reshape(L, B, Ev) @ ow.t() + ob
        torch.testing.assert_close(out, manual, rtol=1e-5, atol=1e-5)
        torch.testing.assert_close(w, p.view(B, H, L, S).mean(1), rtol=1e-5, atol=1e-6)


def test_frozen_batchnorm_matches_torch_batchnorm_in_eval_mode():
    g = torch.Generator().manual_seed(4)
    bn = torch.nn.BatchNorm2d(24).eval()
    fz = tp.FrozenBatchNorm2d(24)
    with torch.no_grad():
        for name in ("weight", "bias", "running_mean"):
            t = torch.randn(24, generator=g)
            getattr(bn, name).copy_(t); getattr(fz, name).copy_(t)
        var = torch.rand(24, generator=g) + 0.1
        bn.running_var.copy_(var); fz.running_var.copy_(var)
    x = torch.randn(3, 24, 9, 7, generator=g)
    torch.testing.assert_close(fz(x), bn(x), rtol=1e-5, atol=1e-5)
    sc, sh = tp.frozen_bn_scale_shift(fz.weight, fz.bias, fz.running_mean, fz.running_var)
    torch.testing.assert_close(x * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1), bn(x), rtol=1e-5, atol=1e-5)


def test_inverse_sigmoid_and_accuracy_closed_forms():
    x = torch.tensor([0.0, 1e-7, 0.25, 0.5, 0.9, 1.0, 1.5, -0.2])
    xc = x.clamp(0, 1)
    want = torch.log(xc.clamp(min=1e-5) / (1 - xc).clamp(min=1e-5))
    torch.testing.assert_close(tp.inverse_sigmoid(x), want)
    torch.testing.assert_close(torch.sigmoid(tp.inverse_sigmoid(torch.tensor([0.2, 0.7]))), torch.tensor([0.2, 0.7]))
    logits = torch.tensor([[0.1, 2.0, -1.0], [3.0, 0.0, 0.5], [0.0, 0.1, 0.2], [1.0, 0.9, 0.8]])
    target = torch.tensor([1, 2, 2, 0])
    top1, top2 = tp.accuracy(logits, target, topk=(1, 2))
    assert float(top1) == 75.0 and float(top2) == 100.0


@pytest.mark.parametrize("name", ["resnet18", "resnet50"])
def test_resnet_standin_matches_the_transformers_resnet(name):
    """The torchvision ResNet stand-in (v1.5: stride on the 3x3, frozen affine BN) against the independent ResNet of
    the `transformers` wheel with the SAME weights mapped across the two naming schemes (randomly initialised here:
    no pretrained checkpoint is fetched)."""
    tr = pytest.importorskip("transformers")
    kind, depths, widths = {"resnet18": ("basic", [2, 2, 2, 2], [64, 128, 256, 512]),
                            "resnet50": ("bottleneck", [3, 4, 6, 3], [256, 512, 1024, 2048])}[name]
    cfg = tr.ResNetConfig(num_channels=3, embedding_size=64, hidden_sizes=widths, depths=depths, layer_type=kind,
                          hidden_act="relu", downsample_in_first_stage=False, downsample_in_bottleneck=False)
    hf = tr.ResNetModel(cfg).eval()
    g = torch.Generator().manual_seed(5)
    with torch.no_grad():
        for n, p in list(hf.named_parameters()) + list(hf.named_buffers()):
            if n.endswith("num_batches_tracked"):
                continue
            if n.endswith("running_var"):
                p.copy_(torch.rand(p.shape, generator=g) + 0.5)
            elif p.dim() == 4:
                p.copy_(torch.randn(p.shape, generator=g) * (2.0 / (p.shape[1] * p.shape[2] * p.shape[3])) ** 0.5)
            else:
                p.copy_(torch.randn(p.shape, generator=g) * 0.1 + (1.0 if n.endswith("normalization.weight") else 0.0))
    ours = tp.IntermediateLayerGetter(tp.ResNet(name, norm_layer=tp.FrozenBatchNorm2d), {"layer4": "0"}).eval()
    hsd = {k: v for k, v in hf.state_dict().items() if not k.endswith("num_batches_tracked")}

    def put(dst_conv, dst_bn, src):
        sd = ours.state_dict()
        sd[dst_conv + ".weight"].copy_(hsd[src + ".convolution.weight"])
        for f in ("weight", "bias", "running_mean", "running_var"):
            sd[dst_bn + "." + f].copy_(hsd[src + ".normalization." + f])

    with torch.no_grad():
        put("conv1", "bn1", "embedder.embedder")
        for s, depth in enumerate(depths):
            for i in range(depth):
                src = f"encoder.stages.{s}.layers.{i}"
                dst = f"layer{s + 1}.{i}"
                n_conv = 2 if kind == "basic" else 3
                for j in range(n_conv):
                    put(f"{dst}.conv{j + 1}", f"{dst}.bn{j + 1}", f"{src}.layer.{j}")
                if f"{src}.shortcut.convolution.weight" in hsd:
                    put(f"{dst}.downsample.0", f"{dst}.downsample.1", f"{src}.shortcut")
    x = torch.randn(2, 3, 65, 97, generator=g)
    with torch.no_grad():
        want = hf(x).last_hidden_state
        got = ours(x)
        got = got["0"] if isinstance(got, dict) else got
    assert got.shape == want.shape
    # fp32 through up to 53 convolutions with un-normalised random weights: compare against the output's magnitude
    assert float((got - want).abs().max()) <= 1e-4 * float(want.abs().max()), (float((got - want).abs().max()), float(want.abs().max()))
