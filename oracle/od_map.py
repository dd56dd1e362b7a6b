"""Loop-form CPU restatement of the AP bookkeeping  --  TEST INFRASTRUCTURE ONLY.

Restates /root/reference/future_od/utils/od_map.py:214-314 as explicit per-sample /
per-class / per-threshold loops (numpy, float32 arithmetic as the reference), which is
what the greedy claim procedure *means*; the reference's own file imports unmodified in
the build container, so this module IS pinned by fixtures generated from it
(tests/golden/make_golden.py, case G9).  Small inputs only (pure-Python loops).
"""
import numpy as np
import torch

SIZE_DELIMS = ((1 / 24) * (1 / 64), (1 / 4) * (1 / 12))        # od_map.py:20-23
TOPK = 50                                                      # od_map.py:255


def _f32(x):
    return np.asarray(x, dtype=np.float32)


def _size_cats(boxes, imsize):
    """od_map.py:152-171 -> bool (..., 4): [all, small, medium, large]."""
    H, W = imsize
    area = (boxes[..., 2] - boxes[..., 0]) * (boxes[..., 3] - boxes[..., 1])
    s0, s1 = np.float32(SIZE_DELIMS[0] * H * W), np.float32(SIZE_DELIMS[1] * H * W)
    small = area <= s0
    medium = (s0 < area) & (area <= s1)
    large = s1 < area
    return np.stack([np.ones_like(small), small, medium, large], axis=-1)


def _iou(p, a):
    """od_map.py:46-70 with the 1e-7 smoothing, one pred box vs one anno box, float32."""
    relu = lambda v: np.maximum(v, np.float32(0))
    area1 = relu(p[2] - p[0]) * relu(p[3] - p[1])
    area2 = relu(a[2] - a[0]) * relu(a[3] - a[1])
    inter = relu(min(p[2], a[2]) - max(p[0], a[0])) * relu(min(p[3], a[3]) - max(p[1], a[1]))
    eps = np.float32(1e-7)
    return (inter + eps) / (area1 + area2 - inter + eps)


def prepare_od_map_stuffs(pred_boxes, pred_scores, anno_boxes, anno_classes, anno_active, imsize):
    """-> confs (T,C,B*K) f32, is_positive (T,C,B*K) bool, size_cats (C,4,B*K) bool, num_annos (C,4) i64."""
    pb, ps = _f32(pred_boxes), _f32(pred_scores)
    ab = _f32(anno_boxes)
    ac, aa = np.asarray(anno_classes), np.asarray(anno_active)
    B, Mp, C = ps.shape
    keep = aa.any(axis=0)
    keep[0] = True                                                  # od_map.py:38-39
    ab, ac, aa = ab[:, keep], ac[:, keep], aa[:, keep]
    N = ab.shape[1]
    K = min(TOPK, Mp)
    thr = torch.arange(0.50, 1.00, 0.05).numpy()                    # od_map.py:248
    T = len(thr)
    confs = np.zeros((T, C, B * K), np.float32)
    is_pos = np.zeros((T, C, B * K), bool)
    sizes = np.zeros((C, 4, B * K), bool)
    num_annos = np.zeros((C, 4), np.int64)
    pred_cats = _size_cats(pb, imsize)
    anno_cats = _size_cats(ab, imsize)
    for b in range(B):
        iou = np.zeros((Mp, N), np.float32)
        for m in range(Mp):
            for n in range(N):
                iou[m, n] = _iou(pb[b, m], ab[b, n])
        for c in range(C):
            avail = (aa[b] == 1) & ((ac[b] == c) | (c == C - 1))    # od_map.py:120-129
            num_annos[c] += (avail[:, None] & anno_cats[b]).sum(0)
            order = torch.from_numpy(ps[b, :, c]).sort(descending=True)[1].numpy()[:K]
            confs[:, c, b * K:(b + 1) * K] = ps[b, order, c]
            sizes[c, :, b * K:(b + 1) * K] = pred_cats[b, order].T
            for t in range(T):
                free = avail.copy()
                for k, m in enumerate(order):
                    cand = np.where(free, iou[m], np.float32(0))
                    n_best = int(cand.argmax())
                    if cand[n_best] >= thr[t]:                       # od_map.py:268-277
                        is_pos[t, c, b * K + k] = True
                        free[n_best] = False
    return confs, is_pos, sizes, num_annos


def average_precision(confs, is_positive, size_categories, num_annos):
    """od_map.py:290-314 for one threshold.  confs (C,P), is_positive (C,P), size_categories (C,4,P),
    num_annos (C,4,iters) -> (C,4) float32."""
    confs = torch.as_tensor(confs)
    C, S, P = size_categories.shape
    ids = confs.argsort(dim=1, descending=True).numpy()
    ap = np.zeros((C, S), np.float32)
    tot = np.asarray(num_annos).sum(axis=2)
    for c in range(C):
        for s in range(S):
            pos = (np.asarray(is_positive)[c] & np.asarray(size_categories)[c, s])[ids[c]]
            cnt = np.asarray(size_categories)[c, s][ids[c]]
            prec = np.cumsum(pos).astype(np.float32) / (np.cumsum(cnt).astype(np.float32) + np.float32(1e-5))
            with np.errstate(divide="ignore", invalid="ignore"):
                ap[c, s] = np.float32((prec * pos).sum(dtype=np.float32)) / np.float32(tot[c, s])
    return ap
