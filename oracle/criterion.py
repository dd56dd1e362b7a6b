"""Functional CPU restatement of the Hungarian-matched set loss  --  TEST INFRASTRUCTURE ONLY.

Follows /root/reference/future_od/models/set_criterion.py and st_detr.py:169-263; the
matcher / focal / GIoU arithmetic is our authoring of the absent ConditionalDETR code
(oracle/thirdparty.py, parity unpinned there).
"""
import torch
import torch.nn.functional as F

from .stdetr import Config
from .thirdparty import (
    accuracy,
    box_cxcywh_to_xyxy,
    generalized_box_iou,
    matcher_cost_matrix,
    sigmoid_focal_loss,
)
from scipy.optimize import linear_sum_assignment


def to_detr_targets(H, W, anno_active, anno_boxes, anno_classes):
    """xyxy pixels -> normalised cxcywh, active rows only.  st_detr.py:237-263."""
    cxcy = 0.5 * (anno_boxes[:, :, 0:2] + anno_boxes[:, :, 2:4])
    wh = anno_boxes[:, :, 2:4] - anno_boxes[:, :, 0:2]
    boxes = torch.cat([cxcy, wh], dim=2) * torch.tensor(
        [1 / W, 1 / H, 1 / W, 1 / H], device=anno_boxes.device).view(1, 1, 4)
    return [{"labels": c[a == 1], "boxes": b[a == 1]}
            for c, b, a in zip(anno_classes, boxes, anno_active)]


@torch.no_grad()
def hungarian_match(cfg: Config, pred_logits, pred_boxes, targets, return_cost=False):
    """-> list of (idx_pred int64 ascending, idx_tgt int64) per sample."""
    tgt_ids = torch.cat([t["labels"] for t in targets])
    tgt_boxes = torch.cat([t["boxes"] for t in targets])
    C = matcher_cost_matrix(pred_logits, pred_boxes, tgt_ids, tgt_boxes,
                            cfg.set_cost_class, cfg.set_cost_bbox, cfg.set_cost_giou).cpu()
    sizes = [len(t["boxes"]) for t in targets]
    out = []
    for b, c in enumerate(C.split(sizes, -1)):
        i, j = linear_sum_assignment(c[b])
        out.append((torch.as_tensor(i, dtype=torch.int64), torch.as_tensor(j, dtype=torch.int64)))
    return (out, C) if return_cost else out


def _src_idx(indices):
    b = torch.cat([torch.full_like(src, i) for i, (src, _) in enumerate(indices)])
    s = torch.cat([src for (src, _) in indices])
    return b, s


def loss_labels(cfg, logits, targets, indices, num_boxes, log=True):
    """set_criterion.py:36-77."""
    B, M, C = logits.shape
    idx = _src_idx(indices)
    tgt_o = torch.cat([t["labels"][J] for t, (_, J) in zip(targets, indices)])
    onehot = torch.zeros((B, M, C), dtype=logits.dtype)
    onehot[idx[0], idx[1], tgt_o] = 1
    out = {"loss_ce": sigmoid_focal_loss(logits, onehot, num_boxes, alpha=cfg.focal_alpha, gamma=2) * M}
    if log:
        out["class_error"] = 100 - accuracy(logits[idx], tgt_o)[0]
    return out


@torch.no_grad()
def loss_cardinality(logits, targets):
    """set_criterion.py:79-91 (note: threshold 0.5 on the *logit*)."""
    lengths = torch.as_tensor([len(t["labels"]) for t in targets])
    card = (logits.max(-1)[0] > 0.5).sum(1)
    return {"cardinality_error": F.l1_loss(card.float(), lengths.float())}


def loss_boxes(boxes, targets, indices, num_boxes):
    """set_criterion.py:93-115."""
    idx = _src_idx(indices)
    src = boxes[idx]
    tgt = torch.cat([t["boxes"][i] for t, (_, i) in zip(targets, indices)], dim=0)
    l1 = F.l1_loss(src, tgt, reduction="none").sum() / num_boxes
    giou = 1 - torch.diag(generalized_box_iou(box_cxcywh_to_xyxy(src), box_cxcywh_to_xyxy(tgt)))
    return {"loss_bbox": l1, "loss_giou": giou.sum() / num_boxes}


def set_criterion(cfg: Config, outputs, targets, world_size=1, num_boxes_global=None,
                  matching_mode="per level", return_indices=False):
    """set_criterion.py:172-217.  `num_boxes_global` stands in for the all_reduce at :189-191."""
    nb = float(sum(len(t["labels"]) for t in targets))
    if num_boxes_global is not None:
        nb = float(num_boxes_global) / world_size
    num_boxes = max(nb, 1.0)
    all_indices = []
    indices = hungarian_match(cfg, outputs["pred_logits"], outputs["pred_boxes"], targets)
    all_indices.append(indices)
    losses = {}
    losses.update(loss_labels(cfg, outputs["pred_logits"], targets, indices, num_boxes))
    losses.update(loss_boxes(outputs["pred_boxes"], targets, indices, num_boxes))
    losses.update(loss_cardinality(outputs["pred_logits"], targets))
    for i, aux in enumerate(outputs.get("aux_outputs", [])):
        if matching_mode == "per level":
            indices = hungarian_match(cfg, aux["pred_logits"], aux["pred_boxes"], targets)
        all_indices.append(indices)
        d = {}
        d.update(loss_labels(cfg, aux["pred_logits"], targets, indices, num_boxes, log=False))
        d.update(loss_boxes(aux["pred_boxes"], targets, indices, num_boxes))
        d.update(loss_cardinality(aux["pred_logits"], targets))
        losses.update({k + f"_{i}": v for k, v in d.items()})
    return (losses, all_indices) if return_indices else losses


def weight_dict(cfg: Config):
    """st_detr.py:67-77."""
    base = {"loss_ce": cfg.cls_loss_coef, "loss_bbox": cfg.bbox_loss_coef, "loss_giou": cfg.giou_loss_coef}
    w = dict(base)
    for i in range(cfg.dec_layers - 1):
        w.update({k + f"_{i}": v for k, v in base.items()})
    return w


def total_loss(cfg: Config, outputs, data, **kw):
    """st_detr.py:169-188 -> (loss, stats, loss_dict)."""
    H, W = data["video"].shape[-2:]
    targets = to_detr_targets(H, W, data["active"], data["boxes"], data["classes"])
    ld = set_criterion(cfg, outputs, targets, **kw)
    wd = weight_dict(cfg)
    loss = sum(ld[k] * wd[k] for k in ld if k in wd)
    stats = {
        "labels": ld["loss_ce"] * wd["loss_ce"],
        "box_l1": ld["loss_bbox"] * wd["loss_bbox"],
        "box_giou": ld["loss_giou"] * wd["loss_giou"],
        "cardinality": ld["cardinality_error"],
        "class_error": ld["class_error"],
    }
    return loss, stats, ld
