"""Functional CPU restatement of the spatiotemporal-DETR forward  --  TEST INFRASTRUCTURE ONLY.

Everything is a pure function of (state_dict, config, inputs) in fp32 torch ops, eval-mode
semantics (dropout = identity; FrozenBN has no train/eval difference).  Autograd flows
through, so `loss.backward()` on the result gives the reference gradients.

The state_dict uses exactly the reference's key schema, so the same
tensors can be loaded into the reference's nn.Modules (tests/golden/make_golden.py) and
into the product model.

Citations are to /root/reference/<path>:<line>.
"""
import math
import re
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F

from .thirdparty import (
    RESNET_SPECS,
    STAGE_WIDTH,
    frozen_bn_scale_shift,
    inverse_sigmoid,
    projection_free_mha,
)


@dataclass
class Config:
    """The knobs runs/_model.py:14-74 hard-codes, made explicit."""

    backbone: str = "resnet50"          # runs/_model.py:20
    hidden_dim: int = 256               # st_detr.py:34
    enc_layers: int = 6                 # st_detr.py:31
    dec_layers: int = 6                 # st_detr.py:32
    nheads: int = 8                     # st_detr.py:36-37
    dim_feedforward: int = 2048         # st_detr.py:33
    num_queries: int = 128              # runs/nusc_spatiotemporal_imu_500ms.py:55
    num_classes: int = 8                # runs/nusc_spatiotemporal_imu_500ms.py:54
    num_images: int = 2                 # runs/_model.py:53
    imu_dim: int = 14                   # runs/_model.py:27
    imu_hidden: int = 128               # runs/_model.py:27
    use_imu: bool = True                # runs/_model.py:26-37
    no_temporal: bool = True            # runs/_model.py:70-72
    first_layer_special_when: str = "always"      # runs/_model.py:65
    image_memory_mode: str = "attend one at a time"  # runs/_model.py:68
    joint_layers: int = 0               # paper.py:180-203 JointEncoder (runs/_model.py:52 passes None); layers w/o IMU attention
    joint_mode: str = "joint"           # "joint" = JointEncoder (paper.py:180-203), "sequential" = JointEncoderSequential (:206-234)
    joint_previmages: int = 0           # sequential only: transformer.py:439-441 previmage_attn blocks per layer
    joint_prevout: bool = False         # sequential only: transformer.py:435-438 prevout_attn
    joint_egodeep: bool = False         # transformer.py:442-447 IMU attention in the joint layers (keys: 1 token sequential, L tokens joint)
    joint_f2f_frames: int = 0           # paper.py:237-277 JointEncoderF2F(hidden_dim, num_frames): > 0 = that encoder, for clips of this many past frames
    dec_slotstates: bool = False        # transformer.py:210-215 decoder layers attend to the previous frame's final queries
    dec_egodeep: bool = False           # transformer.py:217-222 decoder layers attend to the frame's IMU token
    single_frame: bool = False          # paper.py:488-528 SingleFrameCore: no frame is dropped, no joint encoder; its
                                        # SeparateEncoder is called `encoder` (see rename_for_core)
    train_backbone: bool = True         # runs/_model.py:21 (lr_backbone > 0)
    dilation: bool = False              # paper.py:95 replace_stride_with_dilation=[False, False, dilation] (runs/_model.py:22: False)
    concat_imu: bool = False            # paper.py:153-156 SeparateEncoder(concat_imu=True): ego code added to the features, no ego token
    # matcher / loss (st_detr.py:41-51)
    set_cost_class: float = 2.0
    set_cost_bbox: float = 5.0
    set_cost_giou: float = 2.0
    cls_loss_coef: float = 2.0
    bbox_loss_coef: float = 5.0
    giou_loss_coef: float = 2.0
    focal_alpha: float = 0.25

    @property
    def backbone_channels(self):
        return 512 if self.backbone in ("resnet18", "resnet34") else 2048   # paper.py:99


P_CORE = "_model."
P_SEP = P_CORE + "separate_encoder."
P_BB = P_SEP + "backbone."
P_ENC = P_SEP + "transformer.layers."
P_JOINT = P_CORE + "joint_encoder.transformer.layers."
P_DET = P_CORE + "detector."
P_DEC = P_DET + "decoder."


# ======================================================================================
# key schema
# ======================================================================================
F2F_SLOTS = (0, 2, 4, 6, 8, 10, 12)          # positions of the Conv2d modules in the nn.Sequential (ReLUs between)


def f2f_layers(p, n):
    """(Cin, Cout, kernel, dilation) of JointEncoderF2F's seven convolutions, all padding="same" (paper.py:245-261)."""
    return [(n * p, 2 * p, 1, 1), (2 * p, 2 * p, 3, 2), (2 * p, 2 * p, 3, 2), (2 * p, p, 3, 4), (p, p, 3, 8),
            (p, p, 3, 2), (p, p, 7, 1)]


def _lin(spec, key, n_out, n_in):
    spec[key + ".weight"] = ((n_out, n_in), "param")
    spec[key + ".bias"] = ((n_out,), "param")


def _ln(spec, key, d):
    spec[key + ".weight"] = ((d,), "param")
    spec[key + ".bias"] = ((d,), "param")


def resnet_conv_list(name):
    """[(key under body., cin, cout, k, stride, pad, bn key, stage idx)] in execution order."""
    kind, depths, exp = RESNET_SPECS[name]
    convs = [("conv1", 3, 64, 7, 2, 3, "bn1", 0)]
    cin = 64
    for s, (width, depth) in enumerate(zip(STAGE_WIDTH, depths)):
        for i in range(depth):
            stride = 2 if (i == 0 and s > 0) else 1          # (weight shapes only: cfg.dilation changes no shape)
            p = f"layer{s + 1}.{i}."
            if kind == "basic":
                convs.append((p + "conv1", cin, width, 3, stride, 1, p + "bn1", s + 1))
                convs.append((p + "conv2", width, width, 3, 1, 1, p + "bn2", s + 1))
            else:
                convs.append((p + "conv1", cin, width, 1, 1, 0, p + "bn1", s + 1))
                convs.append((p + "conv2", width, width, 3, stride, 1, p + "bn2", s + 1))
                convs.append((p + "conv3", width, width * 4, 1, 1, 0, p + "bn3", s + 1))
            if stride != 1 or cin != width * exp:
                convs.append((p + "downsample.0", cin, width * exp, 1, stride, 0,
                              p + "downsample.1", s + 1))
            cin = width * exp
    return convs


def param_spec(cfg: Config) -> Dict[str, tuple]:
    """name -> (shape, "param" | "frozen" | "buffer").  Mirrors the reference state_dict."""
    D, Dff = cfg.hidden_dim, cfg.dim_feedforward
    spec = {}
    for key, cin, cout, k, _s, _p, bnkey, stage in resnet_conv_list(cfg.backbone):
        trainable = cfg.train_backbone and stage >= 2              # paper.py:102-109
        spec[P_BB + "body." + key + ".weight"] = ((cout, cin, k, k), "param" if trainable else "frozen")
        for b in ("weight", "bias", "running_mean", "running_var"):
            spec[P_BB + "body." + bnkey + "." + b] = ((cout,), "buffer")
    spec[P_BB + "input_proj.weight"] = ((D, cfg.backbone_channels, 1, 1), "param")   # paper.py:112
    spec[P_BB + "input_proj.bias"] = ((D,), "param")
    if cfg.use_imu:
        _lin(spec, P_SEP + "imu_layers.0", cfg.imu_hidden, cfg.imu_dim)             # runs/_model.py:26-30
        _lin(spec, P_SEP + "imu_layers.2", D, cfg.imu_hidden)
    for i in range(cfg.enc_layers):
        p = f"{P_ENC}{i}."
        spec[p + "self_attn.attn.in_proj_weight"] = ((3 * D, D), "param")          # transformer.py:404
        spec[p + "self_attn.attn.in_proj_bias"] = ((3 * D,), "param")
        _lin(spec, p + "self_attn.attn.out_proj", D, D)
        _ln(spec, p + "self_attn.norm1", D)
        _lin(spec, p + "self_attn.mlp.0", Dff, D)
        _lin(spec, p + "self_attn.mlp.3", D, Dff)
        _ln(spec, p + "self_attn.norm2", D)
        if cfg.use_imu:                                                              # transformer.py:442-445
            for n in ("query_content", "query_pos", "key", "value", "fun.out_proj"):
                _lin(spec, p + "egodeep_attend." + n, D, D)
            _ln(spec, p + "egodeep_attend.norm1", D)
            _lin(spec, p + "egodeep_attend.mlp.0", Dff, D)
            _lin(spec, p + "egodeep_attend.mlp.3", D, Dff)
            _ln(spec, p + "egodeep_attend.norm2", D)
            _ln(spec, p + "norm_eda", D)
    def enc_attention(p):                                                           # transformer.py:401-413
        spec[p + "attn.in_proj_weight"] = ((3 * D, D), "param")
        spec[p + "attn.in_proj_bias"] = ((3 * D,), "param")
        _lin(spec, p + "attn.out_proj", D, D)
        _ln(spec, p + "norm1", D)
        _lin(spec, p + "mlp.0", Dff, D)
        _lin(spec, p + "mlp.3", D, Dff)
        _ln(spec, p + "norm2", D)

    if cfg.joint_f2f_frames:                                                        # paper.py:245-261
        pD, n = cfg.hidden_dim, cfg.joint_f2f_frames
        for idx, (cin, cout, k, _dil) in zip(F2F_SLOTS, f2f_layers(pD, n)):
            spec[f"{P_CORE}joint_encoder.f2f_model.{idx}.weight"] = ((cout, cin, k, k), "param")
            spec[f"{P_CORE}joint_encoder.f2f_model.{idx}.bias"] = ((cout,), "param")
    seq = cfg.joint_mode == "sequential"
    for i in range(cfg.joint_layers):                                               # paper.py:180-183, 206
        p = f"{P_JOINT}{i}."
        enc_attention(p + "self_attn.")
        if seq and cfg.joint_prevout:                                               # transformer.py:435-436
            enc_attention(p + "prevout_attn.")
        for j in range(cfg.joint_previmages if seq else 0):                         # transformer.py:439-441
            enc_attention(p + f"previmage_attn.{j}.")
        if cfg.joint_egodeep:                                                       # transformer.py:442-445
            for nm in ("query_content", "query_pos", "key", "value", "fun.out_proj"):
                _lin(spec, p + "egodeep_attend." + nm, D, D)
            _ln(spec, p + "egodeep_attend.norm1", D)
            _lin(spec, p + "egodeep_attend.mlp.0", Dff, D)
            _lin(spec, p + "egodeep_attend.mlp.3", D, Dff)
            _ln(spec, p + "egodeep_attend.norm2", D)
            _ln(spec, p + "norm_eda", D)
    for i in range(cfg.dec_layers):
        p = f"{P_DEC}layers.{i}."
        for n in ("query_content", "query_pos", "key_content", "key_pos", "value", "fun.out_proj"):
            _lin(spec, p + "self_attend." + n, D, D)
        _ln(spec, p + "norm_sa", D)
        for j in range(cfg.num_images):
            q = f"{p}image_attend.{j}."
            names = ["query_content", "key_content", "key_pos", "value", "query_sine", "fun.out_proj"]
            if i == 0:                                                               # transformer.py:321-324
                names.insert(1, "query_pos")
            for n in names:
                _lin(spec, q + n, D, D)
            _ln(spec, f"{p}norm_ia.{j}", D)
        if cfg.dec_slotstates:                                                       # transformer.py:210-213
            for n in ("query_content", "query_pos", "key_content", "key_pos", "value", "fun.out_proj"):
                _lin(spec, p + "slotstates_attend." + n, D, D)
            _ln(spec, p + "norm_ssa", D)
        if cfg.dec_egodeep:                                                          # transformer.py:217-220 (Dff=None)
            for n in ("query_content", "query_pos", "key", "value", "fun.out_proj"):
                _lin(spec, p + "egodeep_attend." + n, D, D)
            _ln(spec, p + "norm_eda", D)
        _lin(spec, p + "feedforward.0", Dff, D)
        _lin(spec, p + "feedforward.3", D, Dff)
        _ln(spec, p + "norm_out", D)
    _ln(spec, P_DEC + "norm", D)
    _lin(spec, P_DEC + "query_scale.layers.0", D, D)                                # transformer.py:328
    _lin(spec, P_DEC + "query_scale.layers.1", D, D)
    _lin(spec, P_DEC + "ref_point_head.layers.0", D, D)                             # transformer.py:329
    _lin(spec, P_DEC + "ref_point_head.layers.1", 2, D)
    _lin(spec, P_DET + "class_embed", cfg.num_classes, D)                           # paper.py:302
    _lin(spec, P_DET + "bbox_embed.layers.0", D, D)                                 # paper.py:303
    _lin(spec, P_DET + "bbox_embed.layers.1", D, D)
    _lin(spec, P_DET + "bbox_embed.layers.2", 4, D)
    spec[P_DET + "query_embed.weight"] = ((cfg.num_queries, D), "param")            # paper.py:304
    return spec


def make_state_dict(cfg: Config, seed: int = 0, dtype=torch.float32) -> Dict[str, torch.Tensor]:
    """Deterministic 'trained-looking' weights: every tensor non-trivial so no term hides.

    Init *rules* follow the reference where they matter for scale (xavier for >1-D
    transformer weights transformer.py:12-15, kaiming fan_out for convs, class bias
    -log(99) paper.py:306-309) but the zero-init of the last bbox layer (paper.py:312-313)
    and the identity FrozenBN buffers are replaced by small random values on purpose.
    """
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for name, (shape, kind) in param_spec(cfg).items():
        leaf = name.rsplit(".", 1)[-1]
        if kind == "buffer":
            if leaf == "weight":
                t = 0.5 + torch.rand(shape, generator=g)
                # damp the residual branch (last BN of every block) so activations stay O(1..10)
                # through 16 randomly initialised blocks; un-damped they reach 1e4 and the encoder's
                # softmax saturates, which makes gradient comparisons ill-conditioned
                if re.search(r"layer\d\.\d+\.(bn3|bn2)\.weight$", name) and (
                        ".bn3." in name or cfg.backbone in ("resnet18", "resnet34")):
                    t = t * 0.25
            elif leaf == "running_var":
                t = 0.5 + torch.rand(shape, generator=g)
            else:
                t = 0.1 * torch.randn(shape, generator=g)
        elif len(shape) == 4:
            fan_out = shape[0] * shape[2] * shape[3]
            t = torch.randn(shape, generator=g) * math.sqrt(2.0 / fan_out)
        elif len(shape) == 2:
            if name.endswith("query_embed.weight"):
                t = torch.randn(shape, generator=g)
            else:
                a = math.sqrt(6.0 / (shape[0] + shape[1]))
                t = (torch.rand(shape, generator=g) * 2 - 1) * a
                if "bbox_embed.layers.2" in name:
                    t = t * 0.1
        else:  # 1-D: biases and LayerNorm affine
            if ".norm" in name and leaf == "weight":
                t = 1.0 + 0.05 * torch.randn(shape, generator=g)
            elif name.endswith("class_embed.bias"):
                t = torch.full(shape, -math.log(99.0)) + 0.05 * torch.randn(shape, generator=g)
            else:
                t = 0.02 * torch.randn(shape, generator=g)
        sd[name] = t.to(dtype)
    return sd


# ======================================================================================
# backbone  (paper.py:83-116 + torchvision ResNet + FrozenBatchNorm2d)
# ======================================================================================
def _bn(sd, key, x):
    scale, shift = frozen_bn_scale_shift(sd[key + ".weight"], sd[key + ".bias"],
                                         sd[key + ".running_mean"], sd[key + ".running_var"])
    return x * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)


def backbone_forward(sd, cfg: Config, images, taps: Optional[dict] = None):
    """images (F,3,H,W) -> (F, hidden, h, w).  paper.py:114-116."""
    body = P_BB + "body."
    kind, depths, exp = RESNET_SPECS[cfg.backbone]

    def conv(x, key, stride, pad, dil=1):
        return F.conv2d(x, sd[body + key + ".weight"], None, stride, pad, dil)

    x = F.relu(_bn(sd, body + "bn1", conv(images, "conv1", 2, 3)))
    x = F.max_pool2d(x, 3, 2, 1)
    if taps is not None:
        taps["stem"] = x
    cin = 64
    for s, (width, depth) in enumerate(zip(STAGE_WIDTH, depths)):
        # torchvision _make_layer(dilate=True), layer4 only (paper.py:95): the stage keeps the resolution, its first
        # block runs at the previous dilation (1), the others with a 3x3 of dilation 2 / padding 2
        dilated = cfg.dilation and s == 3
        for i in range(depth):
            stride = 2 if (i == 0 and s > 0 and not dilated) else 1
            dil = 2 if (dilated and i > 0) else 1
            p = f"layer{s + 1}.{i}."
            idt = x
            if stride != 1 or cin != width * exp:
                idt = _bn(sd, body + p + "downsample.1", conv(x, p + "downsample.0", stride, 0))
            if kind == "basic":
                if dilated:
                    raise NotImplementedError("Dilation > 1 not supported in BasicBlock")     # torchvision's own refusal
                y = F.relu(_bn(sd, body + p + "bn1", conv(x, p + "conv1", stride, 1)))
                y = _bn(sd, body + p + "bn2", conv(y, p + "conv2", 1, 1))
            else:
                y = F.relu(_bn(sd, body + p + "bn1", conv(x, p + "conv1", 1, 0)))
                y = F.relu(_bn(sd, body + p + "bn2", conv(y, p + "conv2", stride, dil, dil)))
                y = _bn(sd, body + p + "bn3", conv(y, p + "conv3", 1, 0))
            x = F.relu(y + idt)
            cin = width * exp
        if taps is not None:
            taps[f"layer{s + 1}"] = x
    return F.conv2d(x, sd[P_BB + "input_proj.weight"], sd[P_BB + "input_proj.bias"])


# ======================================================================================
# positional encodings  (paper.py:32-80, transformer.py:35-48)
# ======================================================================================
def _sine_encode(embed, nfeat, temperature=10000.0):
    """paper.py:75-80: interleaved sin(even)/cos(odd) of embed / T^(2*floor(i/2)/nfeat)."""
    i = torch.arange(nfeat, dtype=torch.float32, device=embed.device)
    dim_t = temperature ** (2 * (i // 2) / nfeat)
    pos = embed[..., None] / dim_t
    return torch.stack((pos[..., 0::2].sin(), pos[..., 1::2].cos()), dim=-1).flatten(-2)


def spatial_pos_table(h, w, c, device="cpu"):
    """(c, h, w) table; identical for every batch element and frame.  paper.py:57-64."""
    ones = torch.ones((h, w), dtype=torch.float32, device=device)
    y = ones.cumsum(0)
    x = ones.cumsum(1)
    eps = 1e-6
    y = _sine_encode(y / (y[-1:, :] + eps) * (2 * math.pi), c // 2)
    x = _sine_encode(x / (x[:, -1:] + eps) * (2 * math.pi), c // 2)
    return torch.cat((y, x), dim=2).permute(2, 0, 1)


def temporal_pos_table(b, l, h, w, c, temporal_offsets=None, extra=0.0, device="cpu"):
    """(b, l, c, h, w).  paper.py:66-73 (unused by runs/: no_temporal=True)."""
    mask = torch.ones((b, l, h, w), dtype=torch.float32, device=device)
    if temporal_offsets is not None:
        t = mask * temporal_offsets[..., None, None] + extra
    else:
        t = mask.cumsum(1)
    t = _sine_encode(t / (t[:, -1:] + 1e-6) * (2 * math.pi), c)
    return t.permute(0, 1, 4, 2, 3)


def query_sine_embed(pos_xy, D=256):
    """pos_xy (M,B,2) in (0,1) -> (M,B,D) ordered (y | x).  transformer.py:35-48."""
    i = torch.arange(D // 2, dtype=torch.float32, device=pos_xy.device)
    dim_t = 10000 ** (2 * (i // 2) / (D // 2))
    px = pos_xy[:, :, 0, None] * (2 * math.pi) / dim_t
    py = pos_xy[:, :, 1, None] * (2 * math.pi) / dim_t
    px = torch.stack((px[:, :, 0::2].sin(), px[:, :, 1::2].cos()), dim=3).flatten(2)
    py = torch.stack((py[:, :, 0::2].sin(), py[:, :, 1::2].cos()), dim=3).flatten(2)
    return torch.cat((py, px), dim=2)


# ======================================================================================
# encoder  (transformer.py:401-512)
# ======================================================================================
def _linear(sd, key, x):
    return F.linear(x, sd[key + ".weight"], sd[key + ".bias"])


def _lnorm(sd, key, x):
    return F.layer_norm(x, x.shape[-1:], sd[key + ".weight"], sd[key + ".bias"], 1e-5)


def stock_mha(sd, key, q_in, k_in, v_in, nheads):
    """torch.nn.MultiheadAttention forward (packed in_proj), seq-first.  transformer.py:404,417."""
    E = q_in.shape[-1]
    w, b = sd[key + ".in_proj_weight"], sd[key + ".in_proj_bias"]
    q = F.linear(q_in, w[:E], b[:E])
    k = F.linear(k_in, w[E:2 * E], b[E:2 * E])
    v = F.linear(v_in, w[2 * E:], b[2 * E:])
    out, _ = projection_free_mha(q, k, v, nheads, sd[key + ".out_proj.weight"],
                                 sd[key + ".out_proj.bias"])
    return out


def egodeep_attention(sd, key, cfg, q_content, q_pos, ego, with_mlp):
    """transformer.py:85-119.  ego (S,B,D) are the key tokens."""
    q = _linear(sd, key + ".query_content", q_content) + _linear(sd, key + ".query_pos", q_pos)
    k = _linear(sd, key + ".key", ego)
    v = _linear(sd, key + ".value", ego)
    out, _ = projection_free_mha(q, k, v, cfg.nheads, sd[key + ".fun.out_proj.weight"],
                                 sd[key + ".fun.out_proj.bias"])
    if with_mlp:
        out = _lnorm(sd, key + ".norm1", out + out)                  # transformer.py:117
        h = F.relu(_linear(sd, key + ".mlp.0", out))
        out = _lnorm(sd, key + ".norm2", out + _linear(sd, key + ".mlp.3", h))
    return out


def encoder_attention(sd, cfg, key, src, q_in, k_in, v_in):
    """EncoderAttention.forward, transformer.py:415-419."""
    a = stock_mha(sd, key + ".attn", q_in, k_in, v_in, cfg.nheads)
    x = _lnorm(sd, key + ".norm1", src + a)                          # transformer.py:417
    h = F.relu(_linear(sd, key + ".mlp.0", x))
    return _lnorm(sd, key + ".norm2", x + _linear(sd, key + ".mlp.3", h))


def encoder_layer(sd, cfg, i, x, pos, ego, prefix=None, prevout=None, memory=None):
    """x,pos (N,Bf,D); ego (S,Bf,D) or None; prevout (N,Bf,D) or None; memory: list of (N,Bf,D), most recent
    first, or None.  transformer.py:449-487."""
    p = f"{prefix or P_ENC}{i}."
    x = encoder_attention(sd, cfg, p + "self_attn", x, x + pos, x + pos, x)
    if prevout is not None and (p + "prevout_attn.attn.in_proj_weight") in sd:        # transformer.py:463-469
        x = encoder_attention(sd, cfg, p + "prevout_attn", x, x + pos, prevout + pos, prevout)
    if memory is not None:                                                            # transformer.py:470-478 (zip)
        for j, prev in enumerate(memory):
            if (p + f"previmage_attn.{j}.attn.in_proj_weight") not in sd:
                break
            x = encoder_attention(sd, cfg, p + f"previmage_attn.{j}", x, x + pos, prev + pos, prev)
    if ego is not None and (p + "egodeep_attend.key.weight") not in sd:
        ego = None                                                    # layer built with use_egodeep=False
    if ego is not None and cfg.use_imu:
        e = egodeep_attention(sd, p + "egodeep_attend", cfg, x, pos, ego, with_mlp=True)
        x = _lnorm(sd, p + "norm_eda", x + e)                         # transformer.py:485-486
    return x


def separate_encoder(sd, cfg, images, imu, taps=None):
    """images (B,L',3,H,W), imu (B,L',14) -> features (B,L',D,h,w), ego (B,L',D).  paper.py:133-170."""
    B, L = images.shape[:2]
    feat = backbone_forward(sd, cfg, images.flatten(0, 1), taps)
    D, h, w = feat.shape[1:]
    if taps is not None:
        taps["input_proj"] = feat
    ego = None
    if imu is not None and cfg.use_imu:
        ego = _linear(sd, P_SEP + "imu_layers.2", F.relu(_linear(sd, P_SEP + "imu_layers.0", imu)))
    if cfg.concat_imu:                                                 # paper.py:153-156
        feat = feat + ego.reshape(B * L, D, 1, 1)
        ego = None
    if cfg.enc_layers > 0:
        pos = spatial_pos_table(h, w, D, feat.device).flatten(1).t()[:, None, :].expand(-1, B * L, -1)
        x = feat.flatten(2).permute(2, 0, 1)                           # (h w) (b l) c
        e = ego.reshape(1, B * L, D) if ego is not None else None
        for i in range(cfg.enc_layers):
            x = encoder_layer(sd, cfg, i, x, pos, e)
            if taps is not None:
                taps[f"enc{i}"] = x
        feat = x.permute(1, 2, 0).reshape(B * L, D, h, w)
    return feat.view(B, L, D, h, w), ego


# ======================================================================================
# decoder  (transformer.py:61-82,122-398) and detector (paper.py:331-419)
# ======================================================================================
def slot_to_slot(sd, key, cfg, q_content, q_pos, k_content, k_pos):
    q = _linear(sd, key + ".query_content", q_content) + _linear(sd, key + ".query_pos", q_pos)
    k = _linear(sd, key + ".key_content", k_content) + _linear(sd, key + ".key_pos", k_pos)
    v = _linear(sd, key + ".value", k_content)
    return projection_free_mha(q, k, v, cfg.nheads, sd[key + ".fun.out_proj.weight"],
                               sd[key + ".fun.out_proj.bias"])[0]


def slot_to_image(sd, key, cfg, q_content, q_pos, q_sine, mem, mem_pos, is_first):
    """Conditional cross-attention.  transformer.py:132-181.  Returns (out, head-mean weights)."""
    M, B, D = q_content.shape
    N = mem.shape[0]
    H = cfg.nheads
    v = _linear(sd, key + ".value", mem)
    qc = _linear(sd, key + ".query_content", q_content)
    if is_first:
        qc = qc + _linear(sd, key + ".query_pos", q_pos)
    qs = _linear(sd, key + ".query_sine", q_sine)
    q = torch.cat([qc.view(M, B, H, D // H), qs.view(M, B, H, D // H)], dim=3).view(M, B, 2 * D)
    ks = _linear(sd, key + ".key_pos", mem_pos)                       # transformer.py:159
    kc = _linear(sd, key + ".key_content", mem)
    if is_first:
        kc = kc + ks
    k = torch.cat([kc.view(N, B, H, D // H), ks.view(N, B, H, D // H)], dim=3).view(N, B, 2 * D)
    return projection_free_mha(q, k, v, H, sd[key + ".fun.out_proj.weight"],
                               sd[key + ".fun.out_proj.bias"])


def decoder_layer(sd, cfg, i, x, q_pos, q_sine, mems, mem_poss, is_first, attn_out=None, slotstates=None, ego=None):
    """transformer.py:242-312.  slotstates (M,B,D): the previous frame's final queries; ego (1,B,D)."""
    p = f"{P_DEC}layers.{i}."
    x = _lnorm(sd, p + "norm_sa", x + slot_to_slot(sd, p + "self_attend", cfg, x, q_pos, x, q_pos))
    for j, (mem, mpos) in enumerate(zip(mems, mem_poss)):
        o, w = slot_to_image(sd, f"{p}image_attend.{j}", cfg, x, q_pos, q_sine, mem, mpos, is_first)
        if attn_out is not None:
            attn_out.append(w)
        x = _lnorm(sd, f"{p}norm_ia.{j}", x + o)
    if cfg.dec_slotstates and slotstates is not None:                 # transformer.py:288-298 (key_pos = query_pos, :368)
        x = _lnorm(sd, p + "norm_ssa", x + slot_to_slot(sd, p + "slotstates_attend", cfg, x, q_pos, slotstates, q_pos))
    if cfg.dec_egodeep and ego is not None:                           # transformer.py:300-307
        x = _lnorm(sd, p + "norm_eda", x + egodeep_attention(sd, p + "egodeep_attend", cfg, x, q_pos, ego, with_mlp=False))
    h = F.relu(_linear(sd, p + "feedforward.0", x))
    return _lnorm(sd, p + "norm_out", x + _linear(sd, p + "feedforward.3", h))


def _mlp(sd, key, x, n):
    for j in range(n):
        x = _linear(sd, f"{key}.layers.{j}", x)
        if j < n - 1:
            x = F.relu(x)
    return x


def decoder_forward(sd, cfg, q_content, q_pos, mems, mem_poss, first_layer_special, attn_out=None, slotstates=None,
                    ego=None):
    """-> (hs (layers,B,M,D), reference (B,M,2)).  transformer.py:332-398."""
    ref = _mlp(sd, P_DEC + "ref_point_head", q_pos, 2).sigmoid().transpose(0, 1)   # (B,M,2)
    sine0 = query_sine_embed(ref.transpose(0, 1), cfg.hidden_dim)
    inter = []
    x = q_content
    for i in range(cfg.dec_layers):
        special = (i == 0) and first_layer_special
        q_sine = sine0 if special else _mlp(sd, P_DEC + "query_scale", x, 2) * sine0
        x = decoder_layer(sd, cfg, i, x, q_pos, q_sine, mems, mem_poss, special, attn_out, slotstates, ego)
        inter.append(_lnorm(sd, P_DEC + "norm", x))
    return torch.stack(inter).transpose(1, 2), ref


def detect(sd, cfg, frame_feat, pos, first_frame, state, attn_out=None, ego=None):
    """frame_feat,pos (N,B,D); state = (earlier frames' features, slot states) or a plain list of features;
    ego (1,B,D) or None.  paper.py:352-419."""
    B = frame_feat.shape[1]
    q_pos = sd[P_DET + "query_embed.weight"].unsqueeze(1).repeat(1, B, 1)
    q_content = torch.zeros_like(q_pos)
    slot = None
    if isinstance(state, tuple):
        state, slot = state
    mems = [frame_feat] + (state if state is not None else [])
    if cfg.image_memory_mode == "attend one at a time":
        mposs = [pos for _ in mems]
    else:
        mposs = [pos]
    special = (first_frame and cfg.first_layer_special_when == "first frame") or \
        cfg.first_layer_special_when == "always"
    hs, ref = decoder_forward(sd, cfg, q_content, q_pos, mems, mposs, special, attn_out, slot, ego)
    new_state = mems[: cfg.num_images - 1]
    if cfg.dec_slotstates:                                            # paper.py:396-399
        new_state = (new_state, hs[-1].transpose(0, 1))
    ref_logit = inverse_sigmoid(ref)
    coords = []
    for lvl in range(hs.shape[0]):
        t = _mlp(sd, P_DET + "bbox_embed", hs[lvl], 3)
        t = torch.cat([t[..., :2] + ref_logit, t[..., 2:]], dim=-1)   # paper.py:410
        coords.append(t.sigmoid())
    coords = torch.stack(coords)
    logits = _linear(sd, P_DET + "class_embed", hs)
    out = {"pred_logits": logits[-1], "pred_boxes": coords[-1],
           "aux_outputs": [{"pred_logits": a, "pred_boxes": b}
                           for a, b in zip(logits[:-1], coords[:-1])]}
    return out, new_state


def detector_forward(sd, cfg, features, pos_enc, skip_dead=False, attn_out=None, ego=None):
    """features,pos_enc (B,L',D,h,w).  paper.py:331-350.

    skip_dead=False replays the reference loop over every frame (only the last `out`
    survives); skip_dead=True runs only the steps whose result is used.
    """
    B, L = features.shape[:2]
    if cfg.image_memory_mode == "attend all at once":                 # paper.py:334-339
        f = features.permute(1, 3, 4, 0, 2).flatten(0, 2)
        p = pos_enc.permute(1, 3, 4, 0, 2).flatten(0, 2)
        e = ego.transpose(0, 1) if (ego is not None and cfg.dec_egodeep) else None       # l b c (paper.py:337-338)
        return detect(sd, cfg, f, p, True, None, attn_out, e)[0]
    f = features.flatten(3).permute(1, 3, 0, 2)                       # l (h w) b c
    p = pos_enc.flatten(3).permute(1, 3, 0, 2)
    egos = [ego[:, l][None] if (ego is not None and cfg.dec_egodeep) else None for l in range(L)]   # paper.py:343-346
    if skip_dead and not cfg.dec_slotstates:                          # slot states make every frame's pass live
        state = [f[l] for l in range(L - 2, max(L - 1 - cfg.num_images, -1), -1)]
        return detect(sd, cfg, f[L - 1], p[L - 1], L == 1, state if L > 1 else None, attn_out, egos[L - 1])[0]
    state, out = None, None
    for l in range(L):
        out, state = detect(sd, cfg, f[l], p[l], l == 0, state,
                            attn_out if l == L - 1 else None, egos[l])
    return out


def rename_for_core(sd, cfg):
    """The same tensors under the key names of the reference core that cfg selects: SingleFrameCore calls its
    SeparateEncoder `encoder` (paper.py:499), FuturePredCore `separate_encoder` (paper.py:444)."""
    if not cfg.single_frame:
        return sd
    return {k.replace(P_SEP, P_CORE + "encoder.", 1) if k.startswith(P_SEP) else k: v for k, v in sd.items()}


def core_forward(sd, cfg, images, imu=None, temporal_offsets=None, skip_dead=False, taps=None,
                 attn_out=None):
    """FuturePredCore.forward (paper.py:448-485) or, with cfg.single_frame, SingleFrameCore.forward (:502-528), which
    differs in not dropping the last frame and having no joint encoder.  images (B,L,3,H,W), imu (B,L,14)."""
    if not cfg.single_frame:
        images = images[:, :-1]
        imu = imu[:, :-1] if imu is not None else None
        if temporal_offsets is not None:
            temporal_offsets = temporal_offsets[:, :-1]
    if skip_dead and cfg.image_memory_mode == "attend one at a time" and not cfg.joint_layers and not cfg.dec_slotstates:
        keep = min(cfg.num_images, images.shape[1])
        images = images[:, -keep:]
        imu = imu[:, -keep:] if imu is not None else None
        # the temporal term is normalised by the LAST frame's offset (paper.py:72), so trimming from the front is
        # exact when offsets are given (with frame indices instead -- offsets None -- it would not be)
        assert cfg.no_temporal or temporal_offsets is not None, "skip_dead needs explicit temporal offsets"
        if temporal_offsets is not None:
            temporal_offsets = temporal_offsets[:, -keep:]
    feat, _ego = separate_encoder(sd, cfg, images, imu, taps)
    B, L, D, h, w = feat.shape
    pos = spatial_pos_table(h, w, D, feat.device)[None, None].expand(B, L, -1, -1, -1)
    if not cfg.no_temporal:
        pos = pos + temporal_pos_table(B, L, h, w, D, temporal_offsets, device=feat.device)
    if cfg.joint_f2f_frames:                                           # paper.py:263-277: frames stacked on channels
        assert L == cfg.joint_f2f_frames
        x = feat.reshape(B, L * D, h, w)                               # "b l c h w -> b (l c) h w"
        layers = f2f_layers(D, L)
        for j, (idx, (cin, cout, k, dil)) in enumerate(zip(F2F_SLOTS, layers)):
            x = F.conv2d(x, sd[f"{P_CORE}joint_encoder.f2f_model.{idx}.weight"],
                         sd[f"{P_CORE}joint_encoder.f2f_model.{idx}.bias"], 1, dil * (k // 2), dil)
            if j < len(layers) - 1:
                x = F.relu(x)
        feat = x[:, None]                                              # (B, 1, D, h, w); the detector sees ONE frame
        pos = pos[:, -1:]
    elif cfg.joint_layers and cfg.joint_mode == "sequential":         # paper.py:219-234
        xs = feat.permute(1, 3, 4, 0, 2).flatten(1, 2)                 # l (h w) b c
        ps = pos.permute(1, 3, 4, 0, 2).flatten(1, 2)
        outs, out, memory = [], None, []
        for l in range(L):
            ego_l = _ego[:, l][None] if _ego is not None else None    # (1, B, D): paper.py:221
            x = xs[l]
            for i in range(cfg.joint_layers):                          # TransformerEncoder.forward, :503-511
                x = encoder_layer(sd, cfg, i, x, ps[l], ego_l, prefix=P_JOINT, prevout=out, memory=memory)
            out = x
            memory = [xs[l]] + memory
            outs.append(out)
        feat = torch.stack(outs, 0).view(L, h, w, B, D).permute(3, 0, 4, 1, 2)
    elif cfg.joint_layers:                                             # paper.py:193-198, layers without IMU attention
        x = feat.permute(3, 4, 1, 0, 2).flatten(0, 2)                 # (h w l) b c
        pj = pos.permute(3, 4, 1, 0, 2).flatten(0, 2)
        ej = _ego.transpose(0, 1) if (_ego is not None and cfg.joint_egodeep) else None    # l b c (paper.py:196-197)
        for i in range(cfg.joint_layers):
            x = encoder_layer(sd, cfg, i, x, pj, ej, prefix=P_JOINT)
        feat = x.view(h, w, L, B, D).permute(3, 2, 4, 0, 1)
    return detector_forward(sd, cfg, feat, pos, skip_dead, attn_out, _ego)


def imu_from_data(data, with_speed=True):
    """st_detr.py:88-90,115-116."""
    keys = ["translation", "acceleration", "rotation", "rotation_rate"] + (["speed"] if with_speed else [])
    return torch.cat([data[k] for k in keys], dim=2)


# ======================================================================================
# tracker baseline (paper.py:531-706)
def tracker_future_predictor(pred1, pred2, temporal_offsets=None, dim_extrapolation=None):
    """TrackerFuturePredictor.forward, paper.py:605-646 (with :538-603)."""
    from scipy.optimize import linear_sum_assignment
    b1, l1, b2, l2 = pred1["pred_boxes"], pred1["pred_logits"], pred2["pred_boxes"], pred2["pred_logits"]
    with torch.no_grad():
        cost = 0.5 * torch.cdist(b2[:, :, 0:2], b1[:, :, 0:2], p=2) \
            + 0.5 * torch.cdist(l2.sigmoid(), l1.sigmoid(), p=float("inf"))            # (B, M, N)   :538-544,641
        B, M, N = cost.shape
        mapping = torch.full((B, M), -1, dtype=torch.int64)                               # :546-558
        for b in range(B):
            r, c = linear_sum_assignment(cost[b].numpy())
            mapping[b, torch.as_tensor(r)] = torch.as_tensor(c)
        if temporal_offsets is None:                                                      # :631-637
            f = 1.0
        else:
            f = ((temporal_offsets[:, 2] - temporal_offsets[:, 1])
                 / (temporal_offsets[:, 1] - temporal_offsets[:, 0]))[:, None, None]
        has = mapping != -1                                                               # :560-588
        idx = mapping.clamp(min=0)
        c1 = b1.gather(1, idx[:, :, None].expand(-1, -1, 4)).clone()
        c1[~has] = b2[~has]
        if dim_extrapolation is None:                                                     # :590-603
            dims = b2[..., 2:4]
        elif dim_extrapolation == "linear":
            dims = torch.clamp(b2[..., 2:4] + (b2[..., 2:4] - c1[..., 2:4]) * f, min=0)
        elif dim_extrapolation == "percentual":
            dims = b2[..., 2:4] * (b2[..., 2:4] / c1[..., 2:4]) ** f
        elif dim_extrapolation == "average":
            dims = (b2[..., 2:4] + c1[..., 2:4]) / 2
        else:
            raise ValueError(dim_extrapolation)
        pos = b2[..., 0:2] + (b2[..., 0:2] - c1[..., 0:2]) * f
        cl = l1.gather(1, idx[..., None].expand(-1, -1, l1.shape[-1])).clone()
        cl[~has] = 0.0
        return {"pred_boxes": torch.cat([pos, dims], dim=2), "pred_logits": 0.5 * (l2 + cl)}


def tracker_core_forward(sd, cfg, images, imu=None, temporal_offsets=None, dim_extrapolation=None):
    """TrackerBaselineCore.forward, paper.py:665-706 (cfg.single_frame key names).  One frame: plain single-frame
    detection.  Three frames: all are encoded (:681), the positional encoding -- temporal term included, normalised by
    the LAST frame's offset -- is built for the whole clip (:684-686), the first two frames are detected each on its
    own slice (:693-700) and the tracker extrapolates to the third (:701)."""
    assert cfg.single_frame
    L = images.shape[1]
    if L == 1:
        return core_forward(sd, cfg, images, imu, temporal_offsets)
    assert L == 3
    feat, ego = separate_encoder(sd, cfg, images, imu)
    B, _, D, h, w = feat.shape
    pos = spatial_pos_table(h, w, D, feat.device)[None, None].expand(B, L, -1, -1, -1)
    if not cfg.no_temporal:
        pos = pos + temporal_pos_table(B, L, h, w, D, temporal_offsets, device=feat.device)
    preds = [detector_forward(sd, cfg, feat[:, l:l + 1], pos[:, l:l + 1], False, None,
                              ego[:, l:l + 1] if ego is not None else None) for l in range(2)]
    return tracker_future_predictor(preds[0], preds[1], temporal_offsets, dim_extrapolation)
