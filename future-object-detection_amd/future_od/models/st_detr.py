"""Task wrapper: clip + IMU in, detections + loss + AP bookkeeping out.

Drop-in for reference future_od/models/st_detr.py: `SpatioTemporalDETRArgs`, `SpatioTemporalDETR`
(same constructor, `forward(data, visualize, epoch, distributed)` -> the 5-tuple the reference's
Trainer consumes, `get_stat_idfs`, `_model`, `_criterion`), and `to_detr_targets`.
Order of work in `forward` differs on purpose: the targets are packed (one small host sync)
BEFORE the core is queued, so the only sync inside the step is the matcher's cost-matrix copy.
"""
from dataclasses import dataclass

import torch
import torch.nn as nn

from future_od.models.set_criterion import (SetCriterion, build_matcher, device_matching_enabled, pack_targets,
                                            run_while_matching, take_unrun_while_matching)
from future_od.native import ops
from future_od.utils.od_map import prepare_od_map_stuffs


@dataclass
class SpatioTemporalDETRArgs:
    # General settings
    num_classes: int
    masks: bool = False
    # Optimization
    lr_backbone: float = 1e-5
    lr: float = 1e-4
    weight_decay: float = 1e-4
    max_norm: float = 0.1
    # Backbone
    backbone: str = "resnet50"
    dilation: bool = False
    position_embedding: str = "sine"
    pretrained_backbone: bool = True
    # Transformer settings
    enc_layers: int = 6
    dec_layers: int = 6
    dim_feedforward: int = 2048
    hidden_dim: int = 256
    dropout: float = 0.1
    enc_nheads: int = 8
    nheads: int = 8
    num_queries: int = 300
    pre_norm: bool = False
    # Matcher settings
    set_cost_class: float = 2.0
    set_cost_bbox: float = 5.0
    set_cost_giou: float = 2.0
    # Loss settings
    aux_loss: bool = True
    cls_loss_coef: float = 2.0
    bbox_loss_coef: float = 5.0
    giou_loss_coef: float = 2.0
    focal_alpha: float = 0.25
    # Data settings
    no_imu_speed: bool = False
    encode_offset: bool = False


class SpatioTemporalDETR(nn.Module):
    def __init__(self, args: SpatioTemporalDETRArgs, model, loss_matching_mode="per level"):
        super().__init__()
        self._model = model
        weight_dict = {"loss_ce": args.cls_loss_coef, "loss_bbox": args.bbox_loss_coef,
                       "loss_giou": args.giou_loss_coef}
        if args.aux_loss:
            for i in range(args.dec_layers - 1):
                weight_dict.update({f"{k}_{i}": v for k, v in
                                    (("loss_ce", args.cls_loss_coef), ("loss_bbox", args.bbox_loss_coef),
                                     ("loss_giou", args.giou_loss_coef))})
        self._criterion = SetCriterion(args.num_classes, matcher=build_matcher(args), weight_dict=weight_dict,
                                       focal_alpha=args.focal_alpha, losses=["labels", "boxes", "cardinality"],
                                       matching_mode=loss_matching_mode)
        self._imu_keys = ["translation", "acceleration", "rotation", "rotation_rate"]
        if not args.no_imu_speed:
            self._imu_keys.append("speed")
        self._encode_offset = args.encode_offset
        self._weight_rows = None

    @staticmethod
    def get_stat_idfs():
        return ["labels", "box_l1", "box_giou", "cardinality", "class_error"]

    def forward(self, data=None, visualize=False, epoch=None, distributed=False):
        images = data["video"]
        B, L, _, H, W = images.shape
        # --- host-side preparation first.  With the loader's host copies of the annotations (recursive_to keeps
        # them) this touches no device tensor: no wait on the previous step's queued backward, so the host
        # queues this forward while the GPU is still draining it.  Without them the boolean indexing below reads
        # the counts back from the device (one sync per sample, as in the reference).
        if device_matching_enabled(images.device):
            # targets packed, counted, matched and normalised on the device: the host never sees a count
            targets = None
            packed = ops.pack_targets_dev(data["boxes"].float().contiguous(), data["classes"].contiguous(),
                                          data["active"].contiguous(), H, W)
            # a data-parallel captured step (future_od/graph.py) feeds the normaliser in: it all-reduces the target
            # count BEFORE replaying the graph, so that no collective sits inside the captured region
            num_boxes = data.get("_num_boxes")
            if num_boxes is None:
                num_boxes = self._criterion.device_num_boxes(packed["count"], distributed)
        else:
            anno = data.get("_host_annotations") or data
            targets = to_detr_targets(H=H, W=W, anno_active=anno["active"], anno_boxes=anno["boxes"],
                                      anno_classes=anno["classes"])
            packed = pack_targets(targets, images.device)
            num_boxes = self._criterion.global_num_boxes(targets, images.device, distributed, lazy=True)
        kwargs = {}
        if data.get("translation") is not None:
            kwargs["imu"] = torch.cat([data[k] for k in self._imu_keys], dim=2)
        if self._encode_offset:
            kwargs["temporal_offsets"] = data["temporal_offsets"]
        # --- device work
        outputs, model_moods = self._model(images, **kwargs)
        if not (isinstance(outputs, dict) and outputs["pred_logits"].dim() == 3):
            raise ValueError("cannot interpret output on the format: %s" % type(outputs))
        # post-processing / AP bookkeeping do not depend on the matching: when the matcher parks the stream they are
        # queued ahead of the parked wait and run while the worker thread solves the assignments
        done = []
        post_proc = lambda: done.append(self.post_proc(outputs["pred_logits"][:, None], outputs["pred_boxes"][:, None],
                                                       data, images))
        run_while_matching(post_proc)
        try:
            loss, stats = self.loss(data, outputs, distributed, targets=targets, packed=packed, num_boxes=num_boxes)
        finally:
            unrun = take_unrun_while_matching()
        for fn in unrun:
            fn()
        od_map_stuffs, post = done[0]
        post["moods"] = model_moods
        return post, None, loss, stats, od_map_stuffs

    def loss(self, data, outputs, distributed, targets=None, packed=None, num_boxes=None):
        if targets is None and packed is None:
            H, W = data["video"].shape[-2:]
            targets = to_detr_targets(H=H, W=W, anno_active=data["active"], anno_boxes=data["boxes"],
                                      anno_classes=data["classes"])
        loss_dict = self._criterion(outputs, targets, distributed, packed=packed, num_boxes=num_boxes)
        wd = self._criterion.weight_dict
        table = loss_dict.table
        Lv = table.shape[0]
        if self._weight_rows is None or self._weight_rows.shape[0] != Lv or self._weight_rows.device != table.device:
            rows = []
            for lv in range(Lv):
                sfx = "" if lv == Lv - 1 else f"_{lv}"
                rows.append([wd.get("loss_ce" + sfx, 0.0), wd.get("loss_bbox" + sfx, 0.0),
                             wd.get("loss_giou" + sfx, 0.0)])
            self._weight_rows = torch.tensor(rows, dtype=torch.float32, device=table.device)
        loss = (table[:, :3] * self._weight_rows).sum()        # == sum_k loss_k * weight_k (st_detr.py:180)
        stats = {
            "labels": (loss_dict["loss_ce"] * wd["loss_ce"]).detach(),
            "box_l1": (loss_dict["loss_bbox"] * wd["loss_bbox"]).detach(),
            "box_giou": (loss_dict["loss_giou"] * wd["loss_giou"]).detach(),
            "cardinality": loss_dict["cardinality_error"],
            "class_error": loss_dict["class_error"],
        }
        return loss, stats

    @torch.no_grad()
    def post_proc(self, class_scores, boxes, data, images):
        """class_scores (logits) [B,Lo,M,C], boxes cxcywh in (0,1) [B,Lo,M,4] -> od_map 4-tuple and the
        output dict of the reference (st_detr.py:190-234)."""
        B, L, _, H, W = images.shape
        scores, boxes_px = ops.post_proc(class_scores.detach().float().contiguous(),
                                         boxes.detach().float().contiguous(), H, W)
        if L == boxes.shape[1]:
            idx = data["annotated_frame_idx"]
            frame_scores, frame_boxes = scores[range(B), idx], boxes_px[range(B), idx]
        else:
            assert boxes.shape[1] == 1, "If different #outs than #ins, #outs must be 1"
            frame_scores, frame_boxes = scores[:, 0], boxes_px[:, 0]
        od = prepare_od_map_stuffs(frame_boxes, frame_scores, data["boxes"], data["classes"], data["active"], (H, W))
        return od, {"class_scores": scores[:, :, None, ...], "boxes": boxes_px[:, :, None, ...]}


def to_detr_targets(H, W, anno_active, anno_boxes, anno_classes):
    """Dense xyxy-pixel annotations -> per-sample {"labels" (Nb,), "boxes" (Nb,4) cxcywh in (0,1)} with
    only the active rows (reference st_detr.py:237-263).  Tiny index bookkeeping; stays in torch."""
    x0y0, x1y1 = anno_boxes[:, :, 0:2], anno_boxes[:, :, 2:4]
    norm = torch.tensor([1 / W, 1 / H, 1 / W, 1 / H], device=anno_boxes.device).view(1, 1, 4)
    cxcywh = torch.cat([0.5 * (x0y0 + x1y1), x1y1 - x0y0], dim=2) * norm
    keep = anno_active == 1
    return [{"labels": anno_classes[b][keep[b]], "boxes": cxcywh[b][keep[b]]} for b in range(anno_boxes.shape[0])]
