"""Model cores of the paper architecture on the HIP kernels.

Drop-in for reference future_od/models/paper.py: PositionalEncoder, CDetrBackbone,
SeparateEncoder, CDetrDetectorSpatioTemporal, FuturePredCore with the reference's constructor
arguments and parameter names.  What differs is how a clip flows through them:

  [B,L,3,H,W] f32  --one strided read-->  frame-major NHWC [(l b),H,W,8]  (compute dtype)
     --ResNet + input_proj (future_od.native.backbone, one autograd node)-->  tokens [(l b), N, D]
     --encoder layers (batch-first, positional TABLE [N,D])-->  per-frame memories [B,N,D]
     --decoder over the last `num_images` frames-->  hs [layers,B,M,D] --> logits / boxes (f32)

Frames that cannot influence the output are not computed: with the shipped "attend one at a time"
detector and no slot states only the last `num_images` past frames reach the prediction (reference
paper.py:347-350,399-402), and the reference's decoder passes for earlier frames are
overwritten.  `skip_dead_frames=False` restores the reference's full sweep (same outputs).
"""
import math
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn
from torch import Tensor

from future_od.models.transformer import MLP, TransformerEncoder
from future_od.native import functional as Fn
from future_od.native import ops
from future_od.native.backbone import ConvWeight, ResNetBody, run_backbone


def is_main_process():
    import torch.distributed as dist
    return not (dist.is_available() and dist.is_initialized()) or dist.get_rank() == 0


class PositionalEncoder(nn.Module):
    """DETR sine encodings.  The kernels consume token-major tables (`spatial_table`,
    `temporal_table`); the reference-shaped getters remain for API compatibility."""

    def __init__(self, no_temporal: bool = False, temperature=10000, extra_temporal_offset: float = 0.0):
        super().__init__()
        self._no_temporal = no_temporal
        self._temperature = temperature
        self._extra_temporal_offset = extra_temporal_offset
        self.scale = 2 * math.pi
        self._tables = {}

    def spatial_table(self, h, w, c, dtype, device):
        """[h*w, c]; input independent, so computed once per (shape, dtype, device)."""
        key = (h, w, c, dtype, str(device))
        if key not in self._tables:
            self._tables[key] = ops.posenc_table(h, w, c, dtype, device, float(self._temperature))
        return self._tables[key]

    def temporal_table(self, b, l, c, dtype, device, temporal_offsets=None):
        """[b, l, c] per-frame encoding (reference paper.py:66-73)."""
        offs = None if temporal_offsets is None else temporal_offsets.float().contiguous()
        return ops.posenc_temporal(b, l, c, dtype, device, offs, float(self._extra_temporal_offset),
                                   float(self._temperature))

    def get_spatial_encoding(self, b, c, h, w, device):
        t = self.spatial_table(h, w, c, torch.float32, device)
        return t.t().reshape(1, c, h, w).expand(b, -1, -1, -1)

    def get_spatio_temporal_encoding(self, b, l, c, h, w, device, temporal_offsets):
        enc = self.get_spatial_encoding(1, c, h, w, device)[:, None].expand(b, l, -1, -1, -1)
        if not self._no_temporal:
            enc = enc + self.temporal_table(b, l, c, torch.float32, device, temporal_offsets)[..., None, None]
        return enc


class CDetrBackbone(nn.Module):
    """ResNet with frozen BatchNorm + 1x1 projection to hidden_dim (reference paper.py:83-116)."""

    def __init__(self, name: str, train_backbone: bool, dilation: bool, hidden_dim: int, pretrained=True):
        super().__init__()
        if pretrained and is_main_process():
            print("[future_od] pretrained torchvision weights cannot be fetched here (no torchvision / network): "
                  "the backbone keeps its random init until a checkpoint is loaded.")
        self.body = ResNetBody(name, dilate_layer4=bool(dilation))
        self.num_channels = 512 if name in ("resnet18", "resnet34") else 2048
        for pname, p in self.body.named_parameters():
            if not train_backbone or ("layer2" not in pname and "layer3" not in pname and "layer4" not in pname):
                p.requires_grad_(False)
        self.input_proj = ConvWeight(self.num_channels, hidden_dim, 1, 1, 0, bias=True)
        nn.init.kaiming_uniform_(self.input_proj.weight, a=math.sqrt(5))
        bound = 1 / math.sqrt(self.num_channels)
        nn.init.uniform_(self.input_proj.bias, -bound, bound)

    def forward_clip(self, clip, dtype):
        """clip f32 [B,L,3,H,W] (view) -> NHWC features [(l b), h, w, hidden]."""
        return run_backbone(clip, self.body, self.input_proj, dtype)


class SeparateEncoder(nn.Module):
    """Per-frame backbone + transformer encoder (reference paper.py:119-170)."""

    def __init__(self, backbone: CDetrBackbone, transformer: TransformerEncoder = None,
                 imu_layers: nn.Module = None, concat_imu: bool = False):
        super().__init__()
        self.backbone = backbone
        self.imu_layers = imu_layers
        self.transformer = transformer
        self.concat_imu = concat_imu

    def forward(self, clip, pos_encoder: PositionalEncoder, imu: Tensor = None, dtype=torch.bfloat16):
        """clip [B,L,3,H,W] f32, imu [B,L,I] f32 -> (tokens [(l b), N, D], (h, w), ego [(l b), D] | None)."""
        B, L = clip.shape[:2]
        feat = self.backbone.forward_clip(clip, dtype)                       # [(l b), h, w, D]
        F_, h, w, D = feat.shape
        tokens = feat.view(F_, h * w, D)
        ego = None
        if imu is not None and self.imu_layers is not None:
            I = imu.shape[-1]
            # (l b)-ordered, zero-padded to the 16-byte vector width, in the compute dtype
            pad = Fn._pad_to(I, Fn._VEC[dtype])
            sb, sl, si = imu.stride()
            assert si == 1
            x = ops.permute3_cast(imu, dtype, (L, B, pad), (sl, sb, 1), valid2=I).view(L * B, pad)
            lin0, lin2 = self.imu_layers[0], self.imu_layers[2]
            h0 = _linear_padded_in(x, lin0, I, relu=True)
            ego = Fn.linear(h0, lin2.weight, lin2.bias)
        if self.concat_imu:
            # reference paper.py:153-156: the ego-motion code is added to every pixel of its frame's features and is
            # then NOT handed on as a token ("concat" in name only)
            if ego is None:
                raise ValueError("concat_imu needs imu data and imu_layers")
            tokens = tokens + ego.unsqueeze(1)
            ego = None
        if self.transformer:
            pos = pos_encoder.spatial_table(h, w, D, dtype, feat.device)
            tokens = self.transformer(tokens, pos, ego)
        return tokens, (h, w), ego


class _PaddedInLinear(torch.autograd.Function):
    """Linear whose in_features (I) is not a multiple of the vector width: the input arrives already
    zero-padded to Ip columns and the weight is padded on the fly; dW is cut back to [N, I]."""

    @staticmethod
    def forward(ctx, x, weight, bias, I, relu):
        N = weight.shape[0]
        Ip = x.shape[-1]
        wd = weight.detach()

        def build():
            out = torch.empty((N, Ip), dtype=x.dtype, device=wd.device)          # columns >= I zero
            return out, [Fn._Job(wd, out, (1, N, Ip), (0, wd.stride(0), wd.stride(1)), valid2=I)]
        wp = Fn.PREP.get(Fn._wkey(weight, f"lin_pad{Ip}", x.dtype), [weight], build)
        y = ops.gemm_nt(x, wp, shift=bias, relu=relu)
        ctx.save_for_backward(x, y)
        ctx.meta = (I, relu, N, Ip)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y = ctx.saved_tensors
        I, relu, N, Ip = ctx.meta
        g = dy.contiguous()
        if relu:
            g = ops.eltwise(Fn.L.EW_RELU_MASK, g, y)
        dw = Fn.zeros_f32((N, Ip), x.device)
        db = Fn.zeros_f32((N,), x.device)
        ops.gemm_tn_acc(g, x, dw, colsum=db, zeroed=True)
        return None, dw[:, :I], db, None, None


def _linear_padded_in(x, lin, I, relu):
    return _PaddedInLinear.apply(x, lin.weight, lin.bias, I, relu)


class CDetrDetectorSpatioTemporal(nn.Module):
    """Query decoder over the current and previous frames' memories (reference paper.py:280-429)."""

    def __init__(self, decoder, num_classes: int, hidden_dim: int, first_layer_special_when, num_queries=300,
                 aux_loss=True, image_memory_mode="attend one at a time"):
        super().__init__()
        self.decoder = decoder
        self.aux_loss = aux_loss
        self.class_embed = nn.Linear(hidden_dim, num_classes)
        self.bbox_embed = MLP(hidden_dim, hidden_dim, 4, 3)
        self.query_embed = nn.Embedding(num_queries, hidden_dim)
        prior_prob = 0.01
        self.class_embed.bias.data = torch.ones(num_classes) * (-math.log((1 - prior_prob) / prior_prob))
        nn.init.constant_(self.bbox_embed.layers[-1].weight.data, 0)
        nn.init.constant_(self.bbox_embed.layers[-1].bias.data, 0)
        assert first_layer_special_when in ("first frame", "always", "never")
        self._first_layer_special_when = first_layer_special_when
        assert image_memory_mode in ("attend one at a time", "attend all at once")
        self.image_memory_mode = image_memory_mode
        self.num_images = len(self.decoder.layers[0].image_attend)
        assert all(self.num_images == len(layer.image_attend) for layer in self.decoder.layers)
        self.use_slotstates = self.decoder.layers[0].slotstates_attend is not None
        assert all(self.use_slotstates == (layer.slotstates_attend is not None) for layer in self.decoder.layers)
        self.use_egodeep = self.decoder.layers[0].egodeep_attend is not None

    def frames_needed(self, L):
        """Past frames that can reach the output: all of them when they form one memory or when slot states carry
        every frame's queries forward, else the last num_images."""
        if self.image_memory_mode == "attend all at once" or self.use_slotstates:
            return L
        return min(self.num_images, L)

    def forward_recurrent(self, frame_tokens: List[Tensor], pos_of_frame, egodeep=None):
        """The reference's sweep over every past frame (paper.py:340-350) when it is live: the final queries of
        frame l are the slot states of frame l+1 (paper.py:396-399).  egodeep [L,B,D] or None."""
        out, prev, slot = None, [], None
        for l, x in enumerate(frame_tokens):
            mems = [x] + prev
            out, slot = self.detect(mems, pos_of_frame(l), l == 0, slot,
                                    egodeep[l] if (egodeep is not None and self.use_egodeep) else None)
            prev = mems[: self.num_images - 1]
        return out

    def forward(self, frame_tokens: List[Tensor], pos, num_frames_total: int, egodeep=None):
        """frame_tokens: the LAST frames of the clip, oldest first, each [B,N,D].
        'attend one at a time' (reference paper.py:340-350): `pos` = encoding of the current (last) frame, [N,D]
        or [B,N,D]; equivalent to the reference's sweep over all frames keeping the last result.
        'attend all at once' (paper.py:334-339): ONE memory of all frames, (l h w)-ordered; `pos` = its encoding,
        [L*N,D] or [B,L*N,D]."""
        K = len(frame_tokens)
        if self.image_memory_mode == "attend all at once":
            assert K == num_frames_total, "every past frame is part of the memory"
            mem = torch.cat(frame_tokens, dim=1) if K > 1 else frame_tokens[0]
            return self.detect([mem], pos, True, None, egodeep if self.use_egodeep else None)[0]   # egodeep [B,L,D]
        first_frame = num_frames_total == 1
        mems = [frame_tokens[K - 1 - j] for j in range(min(K, self.num_images))]   # current, then previous
        return self.detect(mems, pos, first_frame, None, egodeep if self.use_egodeep else None)[0]

    def detect(self, mems, pos_table, first_frame=True, slotstates=None, egodeep=None):
        """-> (outputs, final queries [B,M,D] = the next frame's slot states)."""
        B, N, D = mems[0].shape
        dtype = mems[0].dtype
        M = self.query_embed.weight.shape[0]
        qpos = Fn.cast_ad(self.query_embed.weight, dtype)                         # [M, D]
        x0 = torch.zeros((B, M, D), dtype=dtype, device=qpos.device)
        special = (first_frame and self._first_layer_special_when == "first frame") or \
            self._first_layer_special_when == "always"
        hs, ref = self.decoder(x0, qpos, mems, [pos_table] * len(mems), first_layer_special=special,
                               slotstates=slotstates, egodeep=egodeep)
        Lv = hs.shape[0]
        logits = Fn.linear(hs, self.class_embed.weight, self.class_embed.bias, out_f32=True)      # [Lv,B,M,C] f32
        t = self.bbox_embed(hs)                                                                    # [Lv,B,M,4]
        boxes = Fn.BoxFinishFn.apply(t.view(Lv, B * M, 4), ref, Lv).view(Lv, B, M, 4)             # f32
        out = {"pred_logits": logits[-1], "pred_boxes": boxes[-1]}
        if self.aux_loss:
            out["aux_outputs"] = [{"pred_logits": a, "pred_boxes": b} for a, b in zip(logits[:-1], boxes[:-1])]
        out["_stacked"] = (logits, boxes)          # all levels, final level last (private fast path)
        return out, hs[-1]


class JointEncoder(nn.Module):
    """Self-attention over the tokens of ALL past frames at once (reference paper.py:180-203).  The reference orders
    them (h w l); every operation of an encoder layer is equivariant under token permutations, so they stay
    frame-major here ([B, L*N, D], the layout the detector's 'attend all at once' memory uses too)."""

    def __init__(self, transformer: TransformerEncoder):
        super().__init__()
        self.transformer = transformer

    def forward(self, tokens, pos, egodeep=None):
        """tokens [B, L*N, D]; pos [L*N, D] table or [B, L*N, D]; egodeep [B, L, D] (the IMU tokens of all frames
        are the keys of the layers' IMU attention, reference paper.py:196-198) or None."""
        return self.transformer(tokens, pos, egodeep)


class JointEncoderSequential(nn.Module):
    """Frame-by-frame joint encoding (reference paper.py:206-234): frame l runs through the encoder layers with
    cross-attention onto the PREVIOUS frame's encoded output (`prevout_attn`) and onto the raw features of earlier
    frames, most recent first (`previmage_attn`), plus that frame's IMU token.  Frames stay [B, N, D] slices of
    the frame-major token tensor; no (h w)-first rearrangement exists here."""

    def __init__(self, transformer: TransformerEncoder):
        super().__init__()
        self.transformer = transformer

    def forward(self, frames, pos_of_frame, egodeep=None):
        """frames: list of [B, N, D]; pos_of_frame(l) -> [N, D] table or [B, N, D]; egodeep [L, B, D] or None."""
        outs, out, memory = [], None, []
        for l, x in enumerate(frames):
            out = self.transformer(x, pos_of_frame(l), None if egodeep is None else egodeep[l], prevout=out,
                                   memory=memory)
            memory = [x] + memory
            outs.append(out)
        return outs


class JointEncoderF2F(nn.Module):
    """The F2F-style baseline encoder (reference paper.py:237-277; https://arxiv.org/abs/1803.11496): the past frames'
    feature maps stacked on the channel axis run through seven convolutions (1x1, then 3x3 with dilations 2, 2, 4, 8,
    2, then 7x7), ReLU between them, and come out as ONE frame.  Same parameter names as the reference's
    `nn.Sequential` (`f2f_model.{0,2,...,12}.{weight,bias}`); activations stay NHWC = the token layout [B, h*w, D]."""

    LAYERS = ((1, 1), (3, 2), (3, 2), (3, 4), (3, 8), (3, 2), (7, 1))          # (kernel, dilation)

    def __init__(self, hidden_dim, num_frames):
        super().__init__()
        from future_od.native.backbone import ConvWeight
        p, n = hidden_dim, num_frames
        widths = ((n * p, 2 * p), (2 * p, 2 * p), (2 * p, 2 * p), (2 * p, p), (p, p), (p, p), (p, p))
        mods = []
        for (cin, cout), (k, _d) in zip(widths, self.LAYERS):
            mods += [ConvWeight(cin, cout, k, 1, k // 2, bias=True), nn.ReLU()]
        self.f2f_model = nn.Sequential(*mods[:-1])
        self.num_frames = num_frames

    def forward(self, frames, h, w):
        """frames: list of `num_frames` tensors [B, h*w, D], oldest first -> [B, h*w, D]."""
        assert len(frames) == self.num_frames, (len(frames), self.num_frames)
        B, N, D = frames[0].shape
        x = torch.cat([f.reshape(B, h, w, D) for f in frames], dim=-1)             # channels ordered (l c)
        convs = [m for m in self.f2f_model if not isinstance(m, nn.ReLU)]
        for j, (cw, (k, d)) in enumerate(zip(convs, self.LAYERS)):
            x = Fn.conv2d_same(x, cw.weight, cw.bias, relu=j < len(convs) - 1, dilation=d)
        return x.reshape(B, N, D)


class FuturePredCore(nn.Module):
    """Drop the future frame, encode the past frames, decode the future detections (reference :432-485)."""

    def __init__(self, separate_encoder: SeparateEncoder, joint_encoder, detector: CDetrDetectorSpatioTemporal,
                 pos_encoder: PositionalEncoder):
        super().__init__()
        if joint_encoder is not None and not isinstance(joint_encoder, (JointEncoder, JointEncoderSequential, JointEncoderF2F)):
            raise TypeError(f"unknown joint encoder {type(joint_encoder).__name__}")
        self.separate_encoder = separate_encoder
        self.joint_encoder = joint_encoder
        self.detector = detector
        self.pos_encoder = pos_encoder
        self.compute_dtype = torch.bfloat16
        self.skip_dead_frames = True

    drop_future = True                     # the last frame of the clip is the one to predict, never an input

    def forward(self, images: Tensor, imu: Tensor = None, temporal_offsets: Tensor = None, temporal_table=None):
        """temporal_table: optional f32 [B, past, D] that replaces the temporal term computed from this clip (the
        tracker baseline runs single frames with the term of a three-frame clip)."""
        B, L = images.shape[:2]
        past = L - 1 if self.drop_future else L
        assert past > 0
        all_at_once = self.detector.image_memory_mode == "attend all at once"
        keep = self.detector.frames_needed(past) if (self.skip_dead_frames or all_at_once) else past
        if self.joint_encoder is not None:
            keep = past                        # every frame reaches every other one through the joint self-attention
        clip = images[:, past - keep:past]
        imu_k = imu[:, past - keep:past] if imu is not None else None
        tokens, (h, w), _ego = self.separate_encoder(clip, self.pos_encoder, imu_k, dtype=self.compute_dtype)
        F_, N, D = tokens.shape
        frames = list(tokens.view(keep, B, N, D).unbind(0))
        spatial = self.pos_encoder.spatial_table(h, w, D, self.compute_dtype, tokens.device)      # [N, D]
        # encodings of (a) all kept frames as one token sequence, (b) the current (last) frame alone
        if not self.pos_encoder._no_temporal:
            # + the per-(batch, frame) temporal term (reference paper.py:50-55,66-73), computed over ALL past
            # frames (it is normalised by the last one) and cut to the frames that are kept
            offs = temporal_offsets[:, :past] if temporal_offsets is not None else None
            tt = temporal_table if temporal_table is not None else \
                self.pos_encoder.temporal_table(B, past, D, torch.float32, tokens.device, offs)
            tt = tt[:, past - keep:]
            pos_all = lambda: (spatial.float()[None, None] + tt[:, :, None]).reshape(B, keep * N, D).to(self.compute_dtype)
            pos_at = lambda l: (spatial.float()[None] + tt[:, l, None]).to(self.compute_dtype).contiguous()    # [B, N, D]
        else:
            pos_all = lambda: spatial.repeat(keep, 1)                                             # [keep*N, D]
            pos_at = lambda l: spatial
        pos_last = lambda: pos_at(keep - 1)
        if isinstance(self.joint_encoder, JointEncoderF2F):
            # all past frames -> ONE frame (reference paper.py:273-277); the detector then runs its first-frame pass with
            # the LAST frame's positional encoding and, as the reference's loop does, the FIRST frame's IMU token
            out_tokens = self.joint_encoder(frames, h, w)
            e = _ego.view(keep, B, D)[0] if (_ego is not None and self.detector.use_egodeep) else None
            out = self.detector([out_tokens], pos_last(), num_frames_total=1, egodeep=e)
            return out, [["model happy" for _ in range(L)] for _ in range(B)]
        if isinstance(self.joint_encoder, JointEncoderSequential):
            frames = self.joint_encoder(frames, pos_at, _ego.view(keep, B, D) if _ego is not None else None)
        elif self.joint_encoder is not None:
            pa = pos_all()
            keys = _ego.view(keep, B, D).transpose(0, 1).contiguous() if _ego is not None else None       # [B, keep, D]
            joint = self.joint_encoder(torch.cat(frames, dim=1) if keep > 1 else frames[0], pa, keys)  # [B, keep*N, D]
            frames = list(joint.view(B, keep, N, D).unbind(1))
        ego = _ego.view(keep, B, D) if _ego is not None else None
        if self.detector.use_slotstates and not all_at_once:
            out = self.detector.forward_recurrent(frames, pos_at, ego)
        else:
            e = None if ego is None else (ego.transpose(0, 1).contiguous() if all_at_once else ego[-1])
            out = self.detector(frames, pos_all() if all_at_once else pos_last(), num_frames_total=past, egodeep=e)
        moods = [["model happy" for _ in range(L)] for _ in range(B)]
        return out, moods


class SingleFrameCore(FuturePredCore):
    """Detections for the LAST frame of the clip from the clip itself (reference paper.py:488-528): nothing is dropped,
    there is no joint encoder; with the usual one-frame clips this is plain single-image ConditionalDETR, the paper's
    single-frame baseline.  State-dict keys follow the reference: the SeparateEncoder is called `encoder`."""

    drop_future = False

    def __init__(self, encoder: SeparateEncoder, detector: CDetrDetectorSpatioTemporal, pos_encoder: PositionalEncoder):
        nn.Module.__init__(self)
        self.encoder = encoder
        self.joint_encoder = None
        self.detector = detector
        self.pos_encoder = pos_encoder
        self.compute_dtype = torch.bfloat16
        self.skip_dead_frames = True

    @property
    def separate_encoder(self):
        return self.encoder


class TrackerFuturePredictor(nn.Module):
    """Future detections by assignment between two frames' detections and extrapolation (reference paper.py:531-646):
    cost = 0.5 * centre distance + 0.5 * L-inf distance of class probabilities, one assignment problem per sample
    (host, the same solver as the matcher), then `fod_tracker_extrapolate`.  Evaluation only: no gradients."""

    def __init__(self, dim_extrapolation: str = None):
        super().__init__()
        if dim_extrapolation not in ops.TRACKER_MODES:
            raise ValueError(f"Unknown dim extrapolation: {dim_extrapolation}")
        self._dim_extrapolation = dim_extrapolation

    @torch.no_grad()
    def forward(self, pred1, pred2, temporal_offsets=None):
        b1, l1 = pred1["pred_boxes"].float().contiguous(), pred1["pred_logits"].float().contiguous()
        b2, l2 = pred2["pred_boxes"].float().contiguous(), pred2["pred_logits"].float().contiguous()
        B, M, _ = b2.shape
        N = b1.shape[1]
        cost = ops.tracker_cost(b2, l2, b1, l1)
        mapping = ops.lap_solve_batch_host(cost.cpu(), [N] * B).to(b2.device)             # [B, M]: column or -1
        factor = None
        if temporal_offsets is not None:                       # (t2 - t1) / (t1 - t0), reference :635-637
            t = temporal_offsets.to(device=b2.device, dtype=torch.float32)
            factor = ((t[:, 2] - t[:, 1]) / (t[:, 1] - t[:, 0])).contiguous()
        boxes, logits = ops.tracker_extrapolate(b2, l2, b1, l1, mapping, factor, self._dim_extrapolation)
        return {"pred_boxes": boxes, "pred_logits": logits}


class TrackerBaselineCore(SingleFrameCore):
    """Single-frame detector + tracker extrapolation (reference paper.py:649-706): clips of ONE frame are detected as
    by SingleFrameCore (training); clips of THREE frames yield detections for the first two frames, each on its own,
    and the third frame's detections come from the tracker (evaluation)."""

    def __init__(self, encoder: SeparateEncoder, detector: CDetrDetectorSpatioTemporal, pos_encoder: PositionalEncoder,
                 tracker_future_predictor: TrackerFuturePredictor):
        super().__init__(encoder, detector, pos_encoder)
        self.tracker_future_predictor = tracker_future_predictor

    def forward(self, images: Tensor, imu: Tensor = None, temporal_offsets: Tensor = None):
        B, L = images.shape[:2]
        if L == 1:
            return super().forward(images, imu, temporal_offsets)
        if L != 3:
            raise ValueError("TrackerBaselineCore takes clips of 1 frame (detect) or 3 frames (detect, detect, extrapolate)")
        tt3 = None
        if not self.pos_encoder._no_temporal:
            # the reference builds the encoding of the whole three-frame clip (its temporal term is normalised by
            # the LAST frame's offset, paper.py:66-73) and hands each detector pass its frame's slice (:684-700)
            D = self.detector.query_embed.weight.shape[1]
            tt3 = self.pos_encoder.temporal_table(B, 3, D, torch.float32, images.device, temporal_offsets)
        preds = []
        for l in range(2):                                     # each frame on its own, as a first frame (reference :693-700)
            out, _ = super().forward(images[:, l:l + 1], imu[:, l:l + 1] if imu is not None else None, None,
                                     temporal_table=None if tt3 is None else tt3[:, l:l + 1])
            preds.append(out)
        pred = self.tracker_future_predictor(preds[0], preds[1], temporal_offsets)
        return pred, [["model happy" for _ in range(L)] for _ in range(B)]
