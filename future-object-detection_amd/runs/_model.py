"""`build_model(args, detr_args)`: the architecture the reference ships (reference runs/_model.py:14-83:
ResNet-50, IMU-token encoder, 2-image conditional decoder, spatial-only positional encoding), built
from the HIP-backed modules.  Extra knobs are read from `args` if present and default to the
reference's literals, so the reference's run scripts work unchanged:

    args.compute_dtype : "bf16" (default) | "fp32"      -- arithmetic mode of the kernels
    args.attn_dtype    : "bf16" | "fp8" | "fp8-all"    -- MX-fp8 attention cores (eval-mode math; csrc/attention_fp8.hip)
    args.num_images    : decoder cross-attention blocks (reference literal: 2)
    args.backbone      : "resnet50" (reference literal) | "resnet18" | "resnet34"
    args.skip_dead_frames : True (default) -- do not compute frames that cannot reach the output
"""
import torch
import torch.nn as nn

import future_od.models.transformer as transformer
from future_od.models.paper import (JointEncoder, JointEncoderF2F, JointEncoderSequential, SingleFrameCore, CDetrBackbone, CDetrDetectorSpatioTemporal, FuturePredCore,
                                    PositionalEncoder, SeparateEncoder)
from future_od.models.st_detr import SpatioTemporalDETR, SpatioTemporalDETRArgs
from future_od.native import functional as Fn
from future_od.parallel import FodDataParallel

_DTYPES = {"bf16": torch.bfloat16, "bfloat16": torch.bfloat16, "fp32": torch.float32, "float32": torch.float32,
           torch.bfloat16: torch.bfloat16, torch.float32: torch.float32}


def _joint_encoder(args, detr_args):
    """None as the reference's runs/_model.py:52 passes, or (knobs of this package) `joint_layers` layers of
    paper.py's JointEncoder (`joint_mode="joint"`: all frames' tokens at once) or JointEncoderSequential
    (`joint_mode="sequential"`: frame by frame, `joint_prevout` / `joint_previmages` / `joint_egodeep` choosing
    the cross-attention blocks of its layers, reference transformer.py:422-447)."""
    if getattr(args, "joint_mode", "joint") == "f2f":                      # paper.py:237-277, the F2F baseline encoder
        return JointEncoderF2F(detr_args.hidden_dim, int(getattr(args, "joint_f2f_frames", 2)))
    n = getattr(args, "joint_layers", 0)
    if not n:
        return None
    sequential = getattr(args, "joint_mode", "joint") == "sequential"
    layers = nn.ModuleList(
        transformer.TransformerEncoderLayer(
            D=detr_args.hidden_dim, Nhead=detr_args.enc_nheads, Dff=detr_args.dim_feedforward,
            num_previmages=getattr(args, "joint_previmages", 0) if sequential else 0,
            use_prevout=bool(getattr(args, "joint_prevout", False)) and sequential,
            use_egodeep=bool(getattr(args, "joint_egodeep", False)))
        for _ in range(n))
    enc = transformer.TransformerEncoder(layers=layers)
    return JointEncoderSequential(enc) if sequential else JointEncoder(enc)


def build_model(args, detr_args: SpatioTemporalDETRArgs):
    Fn.PREP.clear()
    num_images = getattr(args, "num_images", 2)
    single = getattr(args, "core", "future_pred") == "single_frame"       # paper.py:488 SingleFrameCore (baseline)
    make_core = (lambda separate_encoder, joint_encoder, **kw: SingleFrameCore(encoder=separate_encoder, **kw)) \
        if single else FuturePredCore
    core = make_core(
        separate_encoder=SeparateEncoder(
            backbone=CDetrBackbone(name=getattr(args, "backbone", "resnet50"),
                                   train_backbone=detr_args.lr_backbone > 0, dilation=False,
                                   hidden_dim=detr_args.hidden_dim, pretrained=detr_args.pretrained_backbone),
            imu_layers=nn.Sequential(nn.Linear(14, 128), nn.ReLU(inplace=True),
                                     nn.Linear(128, detr_args.hidden_dim)),
            transformer=transformer.TransformerEncoder(layers=nn.ModuleList(
                transformer.TransformerEncoderLayer(D=detr_args.hidden_dim, Nhead=detr_args.enc_nheads,
                                                    Dff=detr_args.dim_feedforward, use_egodeep=True)
                for _ in range(detr_args.enc_layers)))),
        joint_encoder=_joint_encoder(args, detr_args),
        detector=CDetrDetectorSpatioTemporal(
            decoder=transformer.TransformerDecoder(
                layers=nn.ModuleList([
                    transformer.TransformerDecoderLayer(D=detr_args.hidden_dim, Nhead=detr_args.nheads,
                                                        Dff=detr_args.dim_feedforward, dropout=0.1,
                                                        num_images=num_images,
                                                        use_slotstates=bool(getattr(args, "dec_slotstates", False)),
                                                        use_egodeep=bool(getattr(args, "dec_egodeep", False)))
                    for _ in range(detr_args.dec_layers)]),
                norm=nn.LayerNorm(detr_args.hidden_dim), return_intermediate=True, D=detr_args.hidden_dim),
            num_classes=detr_args.num_classes, hidden_dim=detr_args.hidden_dim,
            first_layer_special_when="always", num_queries=detr_args.num_queries, aux_loss=True,
            image_memory_mode=getattr(args, "image_memory_mode", "attend one at a time")),
        pos_encoder=PositionalEncoder(no_temporal=getattr(args, "no_temporal", True)))
    core.compute_dtype = _DTYPES[getattr(args, "compute_dtype", "bf16")]
    attn_dtype = getattr(args, "attn_dtype", None)
    if attn_dtype is not None:
        # BASELINE.json configs[4]: "fp8" = MX-fp8 QK^T / PV in the long-sequence attention launches (the encoder's
        # self-attention), "fp8-all" = in every attention launch; a process-wide switch of the kernel layer
        Fn.ATTN_FP8["mode"] = {"bf16": "off", "fp8": "long", "fp8-all": "all"}[attn_dtype]
    core.skip_dead_frames = bool(getattr(args, "skip_dead_frames", True))
    model = SpatioTemporalDETR(args=detr_args, model=core)
    model.to(args.device)
    if getattr(args, "distributed", False):
        model = FodDataParallel(model, device=args.device)
    return model
