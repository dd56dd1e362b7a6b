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
