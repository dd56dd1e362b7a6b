"""ResNet backbone (frozen-BN) + 1x1 input projection as ONE autograd node over the HIP conv kernels.

Forward: f32 NCHW frames -> NHWC compute dtype -> implicit-GEMM convs with the frozen-BN scale
folded into the prepared weights and shift/ReLU/residual fused in the epilogue -> [F, h, w, hidden].
Backward: a hand-written reverse sweep (dgrad with the ReLU mask and the residual-branch gradient
fused in its epilogue, wgrad into fp32 with atomics), stopping at the first trainable block
because stem and layer1 are frozen (reference future_od/models/paper.py:102-109).

Module/parameter names follow torchvision's ResNet so checkpoints load unchanged
(SURVEY.md 8b): body.conv1, body.bn1, body.layer{1..4}.{i}.{conv,bn}{1,2,3}, downsample.{0,1}.
"""
import os

import torch
import torch.nn as nn
from torch.autograd import Function

from . import functional as Fn
from . import ops

RESNET_SPECS = {"resnet18": ("basic", (2, 2, 2, 2), 1),
                "resnet34": ("basic", (3, 4, 6, 3), 1),
                "resnet50": ("bottleneck", (3, 4, 6, 3), 4)}
STAGE_WIDTH = (64, 128, 256, 512)


class FrozenBatchNorm2d(nn.Module):
    """Constant affine y = x*scale + shift from buffers (never updated, never trained)."""

    def __init__(self, n):
        super().__init__()
        self.register_buffer("weight", torch.ones(n))
        self.register_buffer("bias", torch.zeros(n))
        self.register_buffer("running_mean", torch.zeros(n))
        self.register_buffer("running_var", torch.ones(n))
        self.eps = 1e-5

    def _load_from_state_dict(self, state_dict, prefix, *a, **k):
        state_dict.pop(prefix + "num_batches_tracked", None)
        super()._load_from_state_dict(state_dict, prefix, *a, **k)

    def scale_shift(self):
        def make():
            scale = self.weight * (self.running_var + self.eps).rsqrt()
            return scale.contiguous(), (self.bias - self.running_mean * scale).contiguous()
        key = (self.weight._version, self.bias._version, self.running_mean._version, self.running_var._version,
               self.weight.data_ptr())
        hit = getattr(self, "_ss", None)
        if hit is None or hit[0] != key:
            self._ss = (key, make())
        return self._ss[1]


class ConvWeight(nn.Module):
    """Holds an OIHW conv weight (channels_last storage = the kernels' [Cout][kh][kw][Cin] order)."""

    def __init__(self, cin, cout, k, stride, pad, bias=False, dilation=1):
        super().__init__()
        w = torch.empty(cout, cin, k, k)
        nn.init.kaiming_normal_(w, mode="fan_out", nonlinearity="relu")
        self.weight = nn.Parameter(w.contiguous(memory_format=torch.channels_last))
        self.bias = nn.Parameter(torch.zeros(cout)) if bias else None
        # dilation 2 (3x3, stride 1, pad = dilation) is run by BackboneFn as an ordinary 3x3 on the four pixel-parity
        # sub-grids; k / stride / pad describe THAT convolution
        self.k, self.stride, self.pad, self.dilation = k, stride, pad, dilation


class _Block(nn.Module):
    def __init__(self, kind, cin, width, stride, exp, dilation=1):
        super().__init__()
        self.kind = kind
        self.dilation = dilation
        if kind == "basic" and dilation > 1:
            raise NotImplementedError("Dilation > 1 not supported in BasicBlock")       # torchvision's own refusal
        if kind == "basic":
            self.conv1, self.bn1 = ConvWeight(cin, width, 3, stride, 1), FrozenBatchNorm2d(width)
            self.conv2, self.bn2 = ConvWeight(width, width, 3, 1, 1), FrozenBatchNorm2d(width)
        else:
            self.conv1, self.bn1 = ConvWeight(cin, width, 1, 1, 0), FrozenBatchNorm2d(width)
            self.conv2, self.bn2 = ConvWeight(width, width, 3, stride, 1, dilation=dilation), FrozenBatchNorm2d(width)
            self.conv3, self.bn3 = ConvWeight(width, width * 4, 1, 1, 0), FrozenBatchNorm2d(width * 4)
        self.downsample = None
        if stride != 1 or cin != width * exp:
            self.downsample = nn.Sequential(ConvWeight(cin, width * exp, 1, stride, 0),
                                            FrozenBatchNorm2d(width * exp))

    def convs(self):
        """[(ConvWeight, FrozenBN)] main path, then the downsample pair or None."""
        main = [(self.conv1, self.bn1), (self.conv2, self.bn2)]
        if self.kind != "basic":
            main.append((self.conv3, self.bn3))
        ds = (self.downsample[0], self.downsample[1]) if self.downsample is not None else None
        return main, ds


class ResNetBody(nn.Module):
    """conv1/bn1/layer1..4 containers with torchvision's names (what IntermediateLayerGetter keeps)."""

    def __init__(self, name, dilate_layer4=False):
        """dilate_layer4: torchvision's replace_stride_with_dilation=[False, False, True] (reference paper.py:95):
        layer4 keeps layer3's resolution; its first block runs undilated with stride 1, the others with a 3x3 of
        dilation 2 (ResNet._make_layer: previous_dilation for block 0, the stage's dilation after it)."""
        super().__init__()
        kind, depths, exp = RESNET_SPECS[name]
        self.conv1, self.bn1 = ConvWeight(3, 64, 7, 2, 3), FrozenBatchNorm2d(64)
        cin = 64
        for s, (width, depth) in enumerate(zip(STAGE_WIDTH, depths)):
            blocks = []
            dilated = dilate_layer4 and s == 3
            for i in range(depth):
                stride = 2 if (i == 0 and s > 0 and not dilated) else 1
                blocks.append(_Block(kind, cin, width, stride, exp, dilation=2 if (dilated and i > 0) else 1))
                cin = width * exp
            setattr(self, f"layer{s + 1}", nn.Sequential(*blocks))
        self.out_channels = cin

    def blocks(self):
        for s in range(4):
            for blk in getattr(self, f"layer{s + 1}"):
                yield s + 1, blk

    # the dataset's pixel pipeline (reference nu_scenes.py:97-101), applied inside the layout kernel when the clip
    # arrives as raw uint8 frames; plain attributes, not buffers, so the state-dict schema stays the reference's
    PIXEL_MEAN, PIXEL_STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)

    def pixel_norm(self, video):
        if video.dtype != torch.uint8:
            return ()
        c = getattr(self, "_pixel_norm", None)
        if c is None or c[0].device != video.device:
            c = (torch.tensor(self.PIXEL_MEAN, dtype=torch.float32, device=video.device),
                 torch.tensor(self.PIXEL_STD, dtype=torch.float32, device=video.device))
            object.__setattr__(self, "_pixel_norm", c)
        return c


# the frozen stem and its max-pool as one launch (csrc/stem_pool.hip); "0": two launches
FUSED_STEM_POOL = os.environ.get("FOD_FUSED_STEM_POOL", "1") != "0"

# frames per pass of the frozen front (stem .. last frozen block); 0 (default) = all frames at once
FRONT_FRAMES = int(os.environ.get("FOD_FRONT_FRAMES", "0")) or (1 << 30)


def _conv_fwd(x, cw, bn, dtype, relu, residual=None, cin_pad=None, out=None):
    scale, shift = bn.scale_shift()
    w = Fn.prep_conv(cw.weight, dtype, scale, False, cin_pad=cin_pad)
    geom = ops.conv_geom(x.shape, cw.weight.shape[0], cw.k, cw.stride, cw.pad)
    return ops.conv2d_fwd(x, w, geom, shift=shift, residual=residual, relu=relu, work_cin=cw.weight.shape[1],
                          out=out), geom


# "2" (default): every frozen 64-channel bottleneck as one launch of the persistent weight-stationary kernel -- measured at
# 10 x 225 x 400: projection block 196 us against 517 us for its four separate launches, identity blocks 316 against 430
# (profiles/r03v_fused_bottleneck_v4.txt).  "1": only the stage's first block (the policy of the earlier kernel, which
# tied on the identity blocks); "0": none.
FUSED_BOTTLENECK = os.environ.get("FOD_FUSED_BOTTLENECK", "2")


def _fusable(blk, x, dtype, force=False):
    """fod_bottleneck_fused_fwd's domain: bf16, stride-1 bottleneck of width 64 with an identity (Cin 256) or 1x1
    projection (Cin 64) shortcut -- torchvision ResNet-50's layer1.  `force`: the domain only, not the policy."""
    if (FUSED_BOTTLENECK == "0" and not force) or dtype != torch.bfloat16 or blk.kind == "basic":
        return False
    if FUSED_BOTTLENECK == "1" and not force and blk.downsample is None:
        return False
    main, ds = blk.convs()
    c1, c2, c3 = main[0][0], main[1][0], main[2][0]
    cin = x.shape[-1]
    if c1.weight.shape[0] != 64 or c3.weight.shape[0] != 256 or c2.stride != 1 or c1.stride != 1:
        return False
    if ds is None:
        return cin == 256
    return cin == 64 and ds[0].k == 1 and ds[0].stride == 1


def _fused_bottleneck(blk, x, dtype, out=None):
    main, ds = blk.convs()
    ws, bs = [], []
    for cw, bn in main + ([ds] if ds is not None else []):
        scale, shift = bn.scale_shift()
        ws.append(Fn.prep_conv(cw.weight, dtype, scale, False))
        bs.append(shift)
    wd, bd = (ws[3], bs[3]) if ds is not None else (None, None)
    return ops.bottleneck_fused_fwd(x, ws[0], bs[0], ws[1], bs[1], ws[2], bs[2], wd, bd, out=out)


def _parity_split(x):
    """[F,H,W,C] -> [(i j f), ceil(H/2), ceil(W/2), C]: the four pixel-parity sub-grids as extra images (a zero row /
    column where H / W is odd).  A dilation-2 "same" 3x3 on x IS an ordinary pad-1 3x3 on these, and every 1x1 and
    pointwise step of a bottleneck commutes with the rearrangement, so whole dilated blocks run in this layout."""
    F_, H, W, C = x.shape
    h2, w2 = (H + 1) // 2, (W + 1) // 2
    t = (x.new_zeros if (H % 2 or W % 2) else x.new_empty)((2, 2, F_, h2, w2, C))
    for i in range(2):
        for j in range(2):
            sub = x[:, i::2, j::2]
            t[i, j, :, :sub.shape[1], :sub.shape[2]].copy_(sub)
    return t.view(4 * F_, h2, w2, C)


def _parity_merge(t, H, W):
    """Inverse of _parity_split (the pad row / column is dropped)."""
    F4, h2, w2, C = t.shape
    t6 = t.view(2, 2, F4 // 4, h2, w2, C)
    x = t.new_empty((F4 // 4, H, W, C))
    for i in range(2):
        for j in range(2):
            dst = x[:, i::2, j::2]
            dst.copy_(t6[i, j, :, :dst.shape[1], :dst.shape[2]])
    return x


def _zero_parity_pads(t, H, W):
    """The pad row / column of the odd sub-grids must read as the convolution's zero padding."""
    F4, h2, w2, C = t.shape
    t6 = t.view(2, 2, F4 // 4, h2, w2, C)
    if H % 2:
        t6[1, :, :, h2 - 1].zero_()
    if W % 2:
        t6[:, 1, :, :, w2 - 1].zero_()


def _scale7(bn, scale):
    """The frozen-BN scale repeated per tap row ([Cout*7], the row index of the stem weight's permute job)."""
    hit = getattr(bn, "_s7", None)
    if hit is None or hit[0] is not scale:
        bn._s7 = (scale, scale.repeat_interleave(7).contiguous())
    return bn._s7[1]


class BackboneFn(Function):
    @staticmethod
    def forward(ctx, video, body, proj, dtype, *train_weights):
        """video f32 [B,L,3,H,W] (a strided view is fine) -> features NHWC [(l b), h, w, hidden];
        train_weights = trainable conv weights in body.blocks() order, then proj.weight, proj.bias
        (listed only so autograd routes their gradients)."""
        scale1, shift1 = body.bn1.scale_shift()
        w_stem = Fn.prep_stem(body.conv1.weight, dtype, _scale7(body.bn1, scale1))
        norm = body.pixel_norm(video)

        def block_fwd(blk, x, out=None, frozen=False, pads=None):
            """One residual block forward; returns (output, activations, geometries, downsample geometry).
            pads = (H, W) of the full grid when x holds the parity sub-grids of a dilated block."""
            main, ds = blk.convs()
            if frozen and _fusable(blk, x, dtype):
                # a frozen 64-channel bottleneck keeps nothing for backward: one launch, intermediates stay on the CU
                return _fused_bottleneck(blk, x, dtype, out), None, None, None
            idt, ds_geom = x, None
            if ds is not None:
                idt, ds_geom = _conv_fwd(x, ds[0], ds[1], dtype, relu=False)
            acts, geoms = [x], []
            h = x
            for j, (cw, bn) in enumerate(main):
                last = j == len(main) - 1
                h, g = _conv_fwd(h, cw, bn, dtype, relu=True, residual=idt if last else None,
                                 out=out if last else None)
                if pads is not None and j == 0:
                    # the 3x3's input: zeros where the full grid has no pixel.  The backward needs nothing extra --
                    # the ReLU mask of this (zeroed) activation already drops the gradient there
                    _zero_parity_pads(h, *pads)
                acts.append(h)
                geoms.append(g)
            return h, acts, geoms, ds_geom

        blocks = list(body.blocks())
        # The frozen front of the network (stem, max-pool and every leading block without trainable weights: layer1
        # in the reference's configuration, paper.py:102-109) keeps nothing for backward, so it CAN run a few frames
        # at a time (FOD_FRONT_FRAMES) in the hope that a producer's output (46 MB per 900x1600 frame) is still in
        # the 256 MB Infinity Cache when its consumer reads it.  Measured (profiles/r02b): no gain -- conv forward
        # 5.35 ms with all 10 frames per launch, 5.58 / 5.45 ms with 2 / 4 -- so the default is all frames at once.
        n_front = 0
        while (n_front < len(blocks) and blocks[n_front][1].dilation == 1
               and not blocks[n_front][1].convs()[0][0][0].weight.requires_grad):
            n_front += 1
        b_sz, l_sz = video.shape[0], video.shape[1]
        steps = max(1, min(l_sz, FRONT_FRAMES // max(b_sz, 1)))
        x = None
        for l0 in range(0, l_sz, steps):
            part = video[:, l0:l0 + steps]
            xp = ops.clip_to_stem_layout(part, dtype, *norm)
            if FUSED_STEM_POOL and dtype == torch.bfloat16 and w_stem.shape[0] == 64:
                # the stem is frozen (reference paper.py:102-109): conv + BN + ReLU + max-pool in one launch, the
                # full-resolution map stays in LDS
                h = ops.stem_pool_fwd(xp, w_stem, video.shape[-2], video.shape[-1], shift=shift1)
                del xp
            else:
                h = ops.conv_stem_fwd(xp, w_stem, video.shape[-2], video.shape[-1], shift=shift1, relu=True)
                del xp
                h = ops.maxpool3x3s2(h)
            for i in range(n_front):
                dst = None
                if i == n_front - 1 and steps < l_sz:
                    if x is None:
                        cout = blocks[i][1].convs()[0][-1][0].weight.shape[0]
                        x = torch.empty((l_sz * b_sz, h.shape[1], h.shape[2], cout), dtype=dtype, device=video.device)
                    dst = x[l0 * b_sz:(l0 + part.shape[1]) * b_sz]
                h, _, _, _ = block_fwd(blocks[i][1], h, out=dst, frozen=True)
            if steps >= l_sz:
                x = h
            elif n_front == 0:
                if x is None:
                    x = torch.empty((l_sz * b_sz,) + tuple(h.shape[1:]), dtype=dtype, device=video.device)
                x[l0 * b_sz:(l0 + part.shape[1]) * b_sz].copy_(h)
            del h
        tape = []
        dom = None                                       # (H, W) while x holds the parity sub-grids (dilated blocks)
        for _stage, blk in blocks[n_front:]:
            if blk.dilation not in (1, 2):
                raise NotImplementedError(f"bottleneck dilation {blk.dilation}")
            if blk.dilation == 2 and dom is None:
                dom = (x.shape[1], x.shape[2])
                x = _parity_split(x)
            elif blk.dilation == 1 and dom is not None:
                x, dom = _parity_merge(x, *dom), None
            x, acts, geoms, ds_geom = block_fwd(blk, x, pads=dom)
            if blk.convs()[0][0][0].weight.requires_grad:
                tape.append((blk, acts, geoms, ds_geom, dom))
        if dom is not None:
            x = _parity_merge(x, *dom)
        geom_p = ops.conv_geom(x.shape, proj.weight.shape[0], 1, 1, 0)
        wp = Fn.prep_conv(proj.weight, dtype, None, False)
        feat = ops.conv2d_fwd(x, wp, geom_p, shift=proj.bias.detach())
        ctx.body, ctx.proj, ctx.dtype = body, proj, dtype
        ctx.tape, ctx.x_last, ctx.geom_p = tape, x, geom_p
        ctx.train_weights = train_weights
        return feat

    @staticmethod
    def backward(ctx, dfeat):
        dtype, proj = ctx.dtype, ctx.proj
        dev = dfeat.device
        g = dfeat.contiguous()
        grads = {}
        sync = Fn.GRAD_SYNC
        if sync is not None:
            # data parallel: the backbone is the first node of the graph, so it runs last in backward and every
            # transformer gradient is final by now -- their average over ranks travels during the whole sweep below
            sync.flush(Fn.ARENA)

        def wgrad(gy, xin, cw, geom, scale):
            co, ci, kh, kw = cw.weight.shape
            dw = Fn.zeros_f32((co, kh, kw, ci), dev)
            ops.conv2d_wgrad_acc(gy, xin, dw, geom, row_scale=scale, zeroed=True)
            # OIHW view of the [co][kh][kw][ci] buffer (channels_last strides; plain view for 1x1 kernels)
            grads[id(cw.weight)] = dw.view(co, ci, 1, 1) if kh == 1 and kw == 1 else dw.permute(0, 3, 1, 2)

        if proj.weight.requires_grad:
            wgrad(g, ctx.x_last, proj, ctx.geom_p, None)
            db = Fn.zeros_f32((proj.weight.shape[0],), dev)
            ops.colsum_acc(g.view(-1, g.shape[-1]), db)
            grads[id(proj.bias)] = db
        tape = ctx.tape
        if tape:
            # gradient wrt the last block's output, gated by that ReLU (x_last = relu(...))
            g = ops.conv2d_dgrad(g, Fn.prep_conv(proj.weight, dtype, None, True), ctx.geom_p, relu_mask=ctx.x_last)
        gdom = None                                      # the layout g is in, as in forward
        for bi in range(len(tape) - 1, -1, -1):
            blk, acts, geoms, ds_geom, dom = tape[bi]
            if dom is not None and gdom is None:
                g, gdom = _parity_split(g), dom
            elif dom is None and gdom is not None:
                g, gdom = _parity_merge(g, *gdom), None
            main, ds = blk.convs()
            need_dx = bi > 0
            g_out = g                                   # already masked by (block output > 0)
            gcur = g_out
            for j in range(len(main) - 1, -1, -1):
                cw, bn = main[j]
                scale, _ = bn.scale_shift()
                wgrad(gcur, acts[j], cw, geoms[j], scale)
                if j > 0:
                    gcur = ops.conv2d_dgrad(gcur, Fn.prep_conv(cw.weight, dtype, scale, True), geoms[j],
                                            relu_mask=acts[j])
            if ds is not None:
                wgrad(g_out, acts[0], ds[0], ds_geom, ds[1].scale_shift()[0])
            if need_dx:
                cw, bn = main[0]
                sc = bn.scale_shift()[0]
                w1t = Fn.prep_conv(cw.weight, dtype, sc, True)
                if ds is not None and ds[0].k == 1 and ds[0].stride == 2 and cw.stride == 1 and cw.k == 1:
                    # the shortcut is a 1x1 stride-2 convolution: its input gradient lives on the even pixels only.
                    # Main path first (no residual), then the shortcut's ONE parity class accumulated in place -- the
                    # full-resolution, three-quarters-zero gradient tensor is neither written nor read
                    g = ops.conv2d_dgrad(gcur, w1t, geoms[0], relu_mask=acts[0])
                    scd = ds[1].scale_shift()[0]
                    ops.conv2d_dgrad(g_out, Fn.prep_conv(ds[0].weight, dtype, scd, True), ds_geom, residual=g,
                                     relu_mask=acts[0], out=g)
                else:
                    if ds is not None:
                        scd = ds[1].scale_shift()[0]
                        d_idt = ops.conv2d_dgrad(g_out, Fn.prep_conv(ds[0].weight, dtype, scd, True), ds_geom)
                    else:
                        d_idt = g_out
                    g = ops.conv2d_dgrad(gcur, w1t, geoms[0], residual=d_idt, relu_mask=acts[0])
            if sync is not None:
                sync.maybe_flush(Fn.ARENA)        # this block's weight gradients are final (weights are used once)
        ctx.tape = None
        out = [None, None, None, None]
        for w in ctx.train_weights:
            out.append(grads.get(id(w)))
        return tuple(out)


def backbone_trainable(body, proj):
    ws = []
    for _stage, blk in body.blocks():
        main, ds = blk.convs()
        for cw, _bn in main + ([ds] if ds is not None else []):
            if cw.weight.requires_grad:
                ws.append(cw.weight)
    ws += [proj.weight, proj.bias]
    return ws


def run_backbone(video, body, proj, dtype):
    feat = BackboneFn.apply(video, body, proj, dtype, *backbone_trainable(body, proj))
    if Fn.BACKBONE_CUT is not None and feat.requires_grad:
        feat = Fn.BACKBONE_CUT.cut(feat)
    return feat
