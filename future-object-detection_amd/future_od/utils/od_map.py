"""AP bookkeeping (reference future_od/utils/od_map.py): the per-batch greedy TP/FP assignment runs
as one HIP kernel (fod_od_map, one wave per sample x class x IoU threshold) instead of the
reference's 50-step tensor loop; the end-of-epoch aggregation is small host math."""
import numpy as np
import torch

from future_od.native import ops

SIZE_CATEGORY_DELIMITERS = [(1 / 24) * (1 / 64), (1 / 4) * (1 / 12)]   # fractions of H*W
NUM_THRESHOLDS = 10                                                   # IoU 0.50 : 0.05 : 0.95


@torch.no_grad()
def prepare_od_map_stuffs(pred_boxes, pred_class_scores, anno_boxes, anno_classes, anno_active, imsize):
    """pred_boxes (B,M,4) xyxy px, pred_class_scores (B,M,C) incl. the generic class, dense annotations
    (B,N,*) -> confs (T,C,B*K) f32, is_positive (T,C,B*K) bool, size_categories (C,4,B*K) bool,
    num_annos (C,4) i64 with K = min(50, M) (reference od_map.py:214-287)."""
    B, M, C = pred_class_scores.shape
    N = anno_classes.shape[-1]
    return ops.od_map(pred_class_scores.detach().float().contiguous(), pred_boxes.detach().float().contiguous(),
                      anno_boxes.reshape(B, N, 4).float().contiguous(), anno_classes.reshape(B, N).contiguous(),
                      anno_active.reshape(B, N).contiguous(), imsize, T=NUM_THRESHOLDS)


def _get_ap(confs, is_positive, size_categories, num_annos):
    """One IoU threshold: confs (C,P), is_positive (C,P), size_categories (C,S,P), num_annos (C,S,iters)
    -> AP (C,S) as mean precision at the true positives (reference od_map.py:290-314)."""
    C, S, P = size_categories.shape
    order = confs.argsort(dim=1, descending=True).view(C, 1, P).expand(-1, S, -1)
    tp = (is_positive.view(C, 1, P) * size_categories).gather(2, order)
    seen = size_categories.gather(2, order)
    precision = tp.cumsum(dim=2) / (seen.cumsum(dim=2) + 1e-5)
    return (precision * tp).sum(dim=2) / num_annos.sum(dim=2)


def aggregate_mean_average_precision(confs, is_positive, size_categories, num_annos, device):
    """Stacked per-iteration intermediaries -> dict of AP tables (reference od_map.py:317-364).
    Works on any device (the reference's cuda-event timing made it GPU-only)."""
    T = confs.shape[0]
    with torch.no_grad():
        ap = torch.stack([_get_ap(confs[t].to(device), is_positive[t].to(device), size_categories.to(device),
                                  num_annos.to(device)) for t in range(T)]).cpu()
    print("Number of annos for each class and size category is:")
    print(num_annos.sum(dim=2))
    per_class = ap[:, 0:-1, :].numpy()
    return {
        "all": ap[:, 0:-1, :],
        "classavg": np.nanmean(per_class, axis=1),
        "threshavg": np.nanmean(per_class, axis=0),
        "classavg threshavg": np.nanmean(per_class, axis=(0, 1)),
        "generic": ap[:, -1, :],
        "generic threshavg": np.nanmean(ap[:, -1, :].numpy(), axis=0),
    }
