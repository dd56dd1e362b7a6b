"""Our authoring of the third-party symbols on the hot path  --  TEST INFRASTRUCTURE ONLY.

The reference imports nine symbols from the `ConditionalDETR` git submodule
(/root/reference/.gitmodules:1-3, directory empty, pinned commit unknown; upstream
lineage Atten4Vis/ConditionalDETR <- facebookresearch/detr) and the ResNet family from
`torchvision` (requirements.txt:2, not installed in this image).  Neither source is
available, so everything here is written from the published semantics of those
projects and is therefore **parity unpinned** at the third-party boundary.

Reference call sites (all paths relative to /root/reference):
  MultiheadAttention ......... future_od/models/transformer.py:9,64,92,126
  build_matcher .............. future_od/models/st_detr.py:9,65 ; set_criterion.py:182,204
  FrozenBatchNorm2d .......... future_od/models/paper.py:28,97
  sigmoid_focal_loss ......... future_od/models/set_criterion.py:6,63
  box_ops .................... future_od/models/set_criterion.py:7,109-111
  inverse_sigmoid, accuracy .. future_od/models/paper.py:29,406 ; set_criterion.py:8,76
  resnet18/34/50, IntermediateLayerGetter .. future_od/models/paper.py:22-24,94-98,110

`install_standins()` registers these under the absent package names so that
tests/golden/make_golden.py can import and run the reference's own files.
"""
import math
import sys
import types
from collections import OrderedDict

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
from scipy.optimize import linear_sum_assignment


# --------------------------------------------------------------------------------------
# ConditionalDETR.util.box_ops
# --------------------------------------------------------------------------------------
def box_cxcywh_to_xyxy(b):
    cx, cy, w, h = b.unbind(-1)
    return torch.stack([cx - 0.5 * w, cy - 0.5 * h, cx + 0.5 * w, cy + 0.5 * h], dim=-1)


def box_xyxy_to_cxcywh(b):
    x0, y0, x1, y1 = b.unbind(-1)
    return torch.stack([(x0 + x1) / 2, (y0 + y1) / 2, x1 - x0, y1 - y0], dim=-1)


def box_area(b):
    return (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])


def box_iou(a, b):
    """Pairwise IoU of xyxy boxes a (P,4) and b (Q,4) -> (iou (P,Q), union (P,Q))."""
    area_a, area_b = box_area(a), box_area(b)
    top_left = torch.max(a[:, None, :2], b[None, :, :2])
    bot_right = torch.min(a[:, None, 2:], b[None, :, 2:])
    wh = (bot_right - top_left).clamp(min=0)
    inter = wh[..., 0] * wh[..., 1]
    union = area_a[:, None] + area_b[None, :] - inter
    return inter / union, union


def generalized_box_iou(a, b):
    """GIoU = IoU - (hull - union) / hull for well-formed xyxy boxes."""
    assert (a[:, 2:] >= a[:, :2]).all()
    assert (b[:, 2:] >= b[:, :2]).all()
    iou, union = box_iou(a, b)
    hull_tl = torch.min(a[:, None, :2], b[None, :, :2])
    hull_br = torch.max(a[:, None, 2:], b[None, :, 2:])
    hull_wh = (hull_br - hull_tl).clamp(min=0)
    hull = hull_wh[..., 0] * hull_wh[..., 1]
    return iou - (hull - union) / hull


# --------------------------------------------------------------------------------------
# ConditionalDETR.util.misc
# --------------------------------------------------------------------------------------
def inverse_sigmoid(x, eps=1e-5):
    x = x.clamp(min=0, max=1)
    num = x.clamp(min=eps)
    den = (1 - x).clamp(min=eps)
    return torch.log(num / den)


def is_main_process():
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return True
    return dist.get_rank() == 0


@torch.no_grad()
def accuracy(output, target, topk=(1,)):
    """Top-k precision in percent, one value per k."""
    if target.numel() == 0:
        return [torch.zeros([], device=output.device)]
    kmax = max(topk)
    n = target.size(0)
    _, pred = output.topk(kmax, 1, True, True)
    hit = pred.t().eq(target.view(1, -1).expand(kmax, n))
    return [hit[:k].reshape(-1).float().sum(0) * (100.0 / n) for k in topk]


def interpolate(*a, **k):  # only reachable from the unused masks path (set_criterion.py:134)
    return F.interpolate(*a, **k)


def nested_tensor_from_tensor_list(*a, **k):  # unused masks path (set_criterion.py:129)
    raise NotImplementedError("masks path is not on the hot path")


# --------------------------------------------------------------------------------------
# ConditionalDETR.models.segmentation
# --------------------------------------------------------------------------------------
def sigmoid_focal_loss(inputs, targets, num_boxes, alpha: float = 0.25, gamma: float = 2):
    p = inputs.sigmoid()
    ce = F.binary_cross_entropy_with_logits(inputs, targets, reduction="none")
    p_t = p * targets + (1 - p) * (1 - targets)
    loss = ce * ((1 - p_t) ** gamma)
    if alpha >= 0:
        loss = (alpha * targets + (1 - alpha) * (1 - targets)) * loss
    return loss.mean(1).sum() / num_boxes


def dice_loss(inputs, targets, num_boxes):  # unused masks path
    inputs = inputs.sigmoid().flatten(1)
    num = 2 * (inputs * targets).sum(1)
    den = inputs.sum(-1) + targets.sum(-1)
    return (1 - (num + 1) / (den + 1)).sum() / num_boxes


# --------------------------------------------------------------------------------------
# ConditionalDETR.models.backbone.FrozenBatchNorm2d
# --------------------------------------------------------------------------------------
class FrozenBatchNorm2d(nn.Module):
    """BatchNorm2d whose statistics and affine are constant buffers (never trained)."""

    def __init__(self, n):
        super().__init__()
        self.register_buffer("weight", torch.ones(n))
        self.register_buffer("bias", torch.zeros(n))
        self.register_buffer("running_mean", torch.zeros(n))
        self.register_buffer("running_var", torch.ones(n))

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        state_dict.pop(prefix + "num_batches_tracked", None)
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    def forward(self, x):
        scale, shift = frozen_bn_scale_shift(
            self.weight, self.bias, self.running_mean, self.running_var
        )
        return x * scale.reshape(1, -1, 1, 1) + shift.reshape(1, -1, 1, 1)


def frozen_bn_scale_shift(weight, bias, running_mean, running_var, eps=1e-5):
    scale = weight * (running_var + eps).rsqrt()
    return scale, bias - running_mean * scale


# --------------------------------------------------------------------------------------
# ConditionalDETR.models.attention.MultiheadAttention  (projection-free MHA)
# --------------------------------------------------------------------------------------
def projection_free_mha(query, key, value, num_heads, out_w, out_b, dropout_p=0.0, training=False):
    """query (T,B,E), key (S,B,E), value (S,B,Ev) -> (out (T,B,Ev), head-mean weights (B,T,S)).

    q is scaled by (E/heads)^-0.5; heads are contiguous slices of the last dim.
    """
    T, B, E = query.shape
    S = key.shape[0]
    Ev = value.shape[2]
    dh, dv = E // num_heads, Ev // num_heads
    q = (query * (float(dh) ** -0.5)).contiguous().view(T, B * num_heads, dh).transpose(0, 1)
    k = key.contiguous().view(S, B * num_heads, dh).transpose(0, 1)
    v = value.contiguous().view(S, B * num_heads, dv).transpose(0, 1)
    w = torch.softmax(torch.bmm(q, k.transpose(1, 2)), dim=-1)
    w = F.dropout(w, p=dropout_p, training=training)
    o = torch.bmm(w, v).transpose(0, 1).contiguous().view(T, B, Ev)
    o = F.linear(o, out_w, out_b)
    return o, w.view(B, num_heads, T, S).sum(dim=1) / num_heads


class MultiheadAttention(nn.Module):
    def __init__(self, embed_dim, num_heads, dropout=0.0, bias=True, vdim=None, **_unused):
        super().__init__()
        self.embed_dim = embed_dim
        self.num_heads = num_heads
        self.dropout = dropout
        self.vdim = vdim if vdim is not None else embed_dim
        assert embed_dim % num_heads == 0 and self.vdim % num_heads == 0
        self.out_proj = nn.Linear(self.vdim, self.vdim)
        nn.init.constant_(self.out_proj.bias, 0.0)

    def forward(self, query, key, value, key_padding_mask=None, need_weights=True, attn_mask=None):
        assert key_padding_mask is None and attn_mask is None, "masks never used on the hot path"
        return projection_free_mha(
            query, key, value, self.num_heads, self.out_proj.weight, self.out_proj.bias,
            dropout_p=self.dropout, training=self.training,
        )


# --------------------------------------------------------------------------------------
# ConditionalDETR.models.matcher
# --------------------------------------------------------------------------------------
def matcher_cost_matrix(pred_logits, pred_boxes, tgt_ids, tgt_boxes,
                        cost_class=2.0, cost_bbox=5.0, cost_giou=2.0, alpha=0.25, gamma=2.0):
    """(B,M,C) logits, (B,M,4) cxcywh, concatenated targets -> (B, M, sum Nb) cost."""
    B, M = pred_logits.shape[:2]
    p = pred_logits.flatten(0, 1).sigmoid()
    bx = pred_boxes.flatten(0, 1)
    neg = (1 - alpha) * (p ** gamma) * (-(1 - p + 1e-8).log())
    pos = alpha * ((1 - p) ** gamma) * (-(p + 1e-8).log())
    c_cls = pos[:, tgt_ids] - neg[:, tgt_ids]
    c_l1 = torch.cdist(bx, tgt_boxes, p=1)
    c_giou = -generalized_box_iou(box_cxcywh_to_xyxy(bx), box_cxcywh_to_xyxy(tgt_boxes))
    return (cost_bbox * c_l1 + cost_class * c_cls + cost_giou * c_giou).view(B, M, -1)


class HungarianMatcher(nn.Module):
    def __init__(self, cost_class=1.0, cost_bbox=1.0, cost_giou=1.0):
        super().__init__()
        self.cost_class, self.cost_bbox, self.cost_giou = cost_class, cost_bbox, cost_giou

    @torch.no_grad()
    def forward(self, outputs, targets):
        tgt_ids = torch.cat([t["labels"] for t in targets])
        tgt_boxes = torch.cat([t["boxes"] for t in targets])
        C = matcher_cost_matrix(
            outputs["pred_logits"], outputs["pred_boxes"], tgt_ids, tgt_boxes,
            self.cost_class, self.cost_bbox, self.cost_giou,
        ).cpu()
        sizes = [len(t["boxes"]) for t in targets]
        out = []
        for b, c in enumerate(C.split(sizes, -1)):
            i, j = linear_sum_assignment(c[b])
            out.append((torch.as_tensor(i, dtype=torch.int64), torch.as_tensor(j, dtype=torch.int64)))
        return out


def build_matcher(args):
    return HungarianMatcher(args.set_cost_class, args.set_cost_bbox, args.set_cost_giou)


# --------------------------------------------------------------------------------------
# torchvision ResNet (v1.5 bottleneck: stride on the 3x3) + IntermediateLayerGetter
# --------------------------------------------------------------------------------------
RESNET_SPECS = {
    # name: (block kind, blocks per stage, expansion)
    "resnet18": ("basic", (2, 2, 2, 2), 1),
    "resnet34": ("basic", (3, 4, 6, 3), 1),
    "resnet50": ("bottleneck", (3, 4, 6, 3), 4),
}
STAGE_WIDTH = (64, 128, 256, 512)


class _Basic(nn.Module):
    def __init__(self, cin, width, stride, norm, down, dilation=1):
        super().__init__()
        if dilation > 1:                                  # torchvision's BasicBlock refuses it the same way
            raise NotImplementedError("Dilation > 1 not supported in BasicBlock")
        self.conv1 = nn.Conv2d(cin, width, 3, stride, 1, bias=False)
        self.bn1 = norm(width)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(width, width, 3, 1, 1, bias=False)
        self.bn2 = norm(width)
        self.downsample = down

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.bn2(self.conv2(y))
        return self.relu(y + idt)


class _Bottleneck(nn.Module):
    def __init__(self, cin, width, stride, norm, down, dilation=1):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, width, 1, bias=False)
        self.bn1 = norm(width)
        self.conv2 = nn.Conv2d(width, width, 3, stride, dilation, dilation, bias=False)     # padding = dilation
        self.bn2 = norm(width)
        self.conv3 = nn.Conv2d(width, width * 4, 1, bias=False)
        self.bn3 = norm(width * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = down

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.relu(self.bn2(self.conv2(y)))
        y = self.bn3(self.conv3(y))
        return self.relu(y + idt)


class ResNet(nn.Module):
    def __init__(self, name, norm_layer=nn.BatchNorm2d, replace_stride_with_dilation=None,
                 pretrained=False, num_classes=1000, **_unused):
        super().__init__()
        assert not pretrained, "no network: pretrained weights unavailable"
        # torchvision ResNet._make_layer(dilate=True): the stage's stride becomes 1 and its dilation is multiplied
        # by that stride; the FIRST block still runs at the previous dilation, the others at the new one
        dilate = tuple(replace_stride_with_dilation or (False, False, False))
        assert len(dilate) == 3
        dilation = 1
        kind, depths, exp = RESNET_SPECS[name]
        block = _Basic if kind == "basic" else _Bottleneck
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = norm_layer(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        cin = 64
        for s, (width, depth) in enumerate(zip(STAGE_WIDTH, depths)):
            blocks = []
            previous_dilation = dilation
            stage_stride = 2 if s > 0 else 1
            if s > 0 and dilate[s - 1]:
                dilation *= stage_stride
                stage_stride = 1
            for i in range(depth):
                stride = stage_stride if i == 0 else 1
                down = None
                if stride != 1 or cin != width * exp:
                    down = nn.Sequential(nn.Conv2d(cin, width * exp, 1, stride, bias=False),
                                         norm_layer(width * exp))
                blocks.append(block(cin, width, stride, norm_layer, down,
                                    dilation=previous_dilation if i == 0 else dilation))
                cin = width * exp
            setattr(self, f"layer{s + 1}", nn.Sequential(*blocks))
        self.avgpool = nn.AdaptiveAvgPool2d(1)
        self.fc = nn.Linear(cin, num_classes)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")


class IntermediateLayerGetter(nn.ModuleDict):
    """Keeps the children of `model` up to the last requested one; returns the tapped outputs."""

    def __init__(self, model, return_layers):
        want = dict(return_layers)
        kept = OrderedDict()
        for name, child in model.named_children():
            kept[name] = child
            want.pop(name, None)
            if not want:
                break
        super().__init__(kept)
        self.return_layers = dict(return_layers)

    def forward(self, x):
        out = OrderedDict()
        for name, child in self.items():
            x = child(x)
            if name in self.return_layers:
                out[self.return_layers[name]] = x
        return out


# --------------------------------------------------------------------------------------
# registration under the absent package names (used ONLY by tests/golden/make_golden.py)
# --------------------------------------------------------------------------------------
def install_standins():
    me = sys.modules[__name__]

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    if "ConditionalDETR.models.attention" not in sys.modules:
        box_ops = mod("ConditionalDETR.util.box_ops", box_cxcywh_to_xyxy=box_cxcywh_to_xyxy,
                      box_xyxy_to_cxcywh=box_xyxy_to_cxcywh, box_iou=box_iou, box_area=box_area,
                      generalized_box_iou=generalized_box_iou)
        misc = mod("ConditionalDETR.util.misc", inverse_sigmoid=inverse_sigmoid,
                   is_main_process=is_main_process, accuracy=accuracy, interpolate=interpolate,
                   nested_tensor_from_tensor_list=nested_tensor_from_tensor_list)
        util = mod("ConditionalDETR.util", box_ops=box_ops, misc=misc)
        att = mod("ConditionalDETR.models.attention", MultiheadAttention=MultiheadAttention)
        mat = mod("ConditionalDETR.models.matcher", HungarianMatcher=HungarianMatcher,
                  build_matcher=build_matcher)
        bb = mod("ConditionalDETR.models.backbone", FrozenBatchNorm2d=FrozenBatchNorm2d)
        seg = mod("ConditionalDETR.models.segmentation", sigmoid_focal_loss=sigmoid_focal_loss,
                  dice_loss=dice_loss)
        models = mod("ConditionalDETR.models", attention=att, matcher=mat, backbone=bb,
                     segmentation=seg)
        mod("ConditionalDETR", util=util, models=models)
    if "torchvision" not in sys.modules:
        def _factory(name):
            return lambda **kw: ResNet(name, **kw)

        utils = mod("torchvision.models._utils", IntermediateLayerGetter=IntermediateLayerGetter)
        tvm = mod("torchvision.models", _utils=utils, ResNet=ResNet,
                  **{n: _factory(n) for n in RESNET_SPECS})
        mod("torchvision", models=tvm)
    return me
