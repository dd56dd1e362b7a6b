#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/*.npz from the REFERENCE's own source files.

Run in the build container only (needs /root/reference):

    python tests/golden/make_golden.py

What it does: registers oracle.thirdparty under the two package names the reference
imports but does not ship (ConditionalDETR.*, torchvision -- SURVEY.md 8c), imports the
reference's future_od/models/{paper,transformer,st_detr,set_criterion}.py and
future_od/utils/od_map.py unmodified, instantiates its nn.Modules through their own
constructors, loads the deterministic weights of oracle.stdetr.make_state_dict into
them and records outputs / losses / gradients on seeded synthetic inputs.  Only data
(inputs' seeds, expected outputs) is written; no reference source is copied.

The fixtures pin the reference's OWN code.  The third-party arithmetic inside them
(MHA core, matcher, focal, GIoU, ResNet) is oracle.thirdparty = parity unpinned.
"""
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "future-object-detection_amd"))

from oracle import thirdparty  # noqa: E402
from oracle.stdetr import Config, make_state_dict, param_spec, rename_for_core  # noqa: E402

REF = "/root/reference"


def import_reference():
    thirdparty.install_standins()
    sys.modules.setdefault("wandb", types.ModuleType("wandb"))
    # make sure `future_od` resolves to the reference, not to our drop-in package
    for k in [k for k in sys.modules if k == "future_od" or k.startswith("future_od.")]:
        del sys.modules[k]
    sys.path.insert(0, REF)
    import future_od.models.paper as paper
    import future_od.models.set_criterion as set_criterion
    import future_od.models.st_detr as st_detr
    import future_od.models.transformer as transformer
    import future_od.utils.od_map as od_map
    sys.path.remove(REF)
    assert paper.__file__.startswith(REF)
    return paper, transformer, st_detr, set_criterion, od_map


def joint_encoder(cfg: Config, paper, transformer):
    """None (runs/_model.py:52), paper.JointEncoder, or paper.JointEncoderSequential with the reference's own
    TransformerEncoderLayer options (transformer.py:423-447)."""
    if cfg.joint_f2f_frames:
        return paper.JointEncoderF2F(cfg.hidden_dim, cfg.joint_f2f_frames)
    if not cfg.joint_layers:
        return None
    seq = cfg.joint_mode == "sequential"
    enc = transformer.TransformerEncoder(layers=nn.ModuleList(
        transformer.TransformerEncoderLayer(D=cfg.hidden_dim, Nhead=cfg.nheads, Dff=cfg.dim_feedforward,
                                            num_previmages=cfg.joint_previmages if seq else 0,
                                            use_prevout=cfg.joint_prevout and seq,
                                            use_egodeep=cfg.joint_egodeep)
        for _ in range(cfg.joint_layers)))
    return paper.JointEncoderSequential(enc) if seq else paper.JointEncoder(enc)


def build_reference(cfg: Config, paper, transformer, st_detr):
    """The graph of runs/_model.py:14-74 with its literals replaced by cfg."""
    args = st_detr.SpatioTemporalDETRArgs(
        num_classes=cfg.num_classes, num_queries=cfg.num_queries, lr_backbone=1e-4,
        enc_layers=cfg.enc_layers, dec_layers=cfg.dec_layers, dim_feedforward=cfg.dim_feedforward,
        hidden_dim=cfg.hidden_dim, enc_nheads=cfg.nheads, nheads=cfg.nheads,
        pretrained_backbone=False, encode_offset=not cfg.no_temporal)
    enc = transformer.TransformerEncoder(layers=nn.ModuleList(
        transformer.TransformerEncoderLayer(D=cfg.hidden_dim, Nhead=cfg.nheads, Dff=cfg.dim_feedforward,
                                            use_egodeep=cfg.use_imu)
        for _ in range(cfg.enc_layers)))
    sep = paper.SeparateEncoder(
        backbone=paper.CDetrBackbone(name=cfg.backbone, train_backbone=True, dilation=cfg.dilation,
                                     hidden_dim=cfg.hidden_dim, pretrained=False),
        imu_layers=nn.Sequential(nn.Linear(cfg.imu_dim, cfg.imu_hidden), nn.ReLU(inplace=True),
                                 nn.Linear(cfg.imu_hidden, cfg.hidden_dim)) if cfg.use_imu else None,
        transformer=enc, concat_imu=cfg.concat_imu)
    core_kw = dict(encoder=sep) if cfg.single_frame else dict(separate_encoder=sep,
                                                              joint_encoder=joint_encoder(cfg, paper, transformer))
    core = (paper.SingleFrameCore if cfg.single_frame else paper.FuturePredCore)(
        **core_kw,
        detector=paper.CDetrDetectorSpatioTemporal(
            decoder=transformer.TransformerDecoder(
                layers=nn.ModuleList([
                    transformer.TransformerDecoderLayer(D=cfg.hidden_dim, Nhead=cfg.nheads,
                                                        Dff=cfg.dim_feedforward, dropout=0.1,
                                                        num_images=cfg.num_images,
                                                        use_slotstates=cfg.dec_slotstates, use_egodeep=cfg.dec_egodeep)
                    for _ in range(cfg.dec_layers)]),
                norm=nn.LayerNorm(cfg.hidden_dim), return_intermediate=True, D=cfg.hidden_dim),
            num_classes=cfg.num_classes, hidden_dim=cfg.hidden_dim,
            first_layer_special_when=cfg.first_layer_special_when, num_queries=cfg.num_queries,
            aux_loss=True, image_memory_mode=cfg.image_memory_mode),
        pos_encoder=paper.PositionalEncoder(no_temporal=cfg.no_temporal))
    model = st_detr.SpatioTemporalDETR(args=args, model=core)
    return model


def load_weights(model, cfg, seed):
    sd = rename_for_core(make_state_dict(cfg, seed), cfg)
    ref_keys = set(model.state_dict().keys())
    assert ref_keys == set(sd.keys()), (sorted(ref_keys ^ set(sd.keys()))[:20])
    model.load_state_dict(sd)
    model.eval()
    return sd


ONLY = [a for a in sys.argv[1:] if not a.startswith("-")]     # fixture names to (re)generate; default: all


def save(name, **arrays):
    if ONLY and name not in ONLY:
        return
    out = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}  ({os.path.getsize(path) / 1024:.1f} KiB)")


def subsample(t, n=4096, seed=7):
    flat = t.detach().reshape(-1)
    if flat.numel() <= n:
        return torch.arange(flat.numel()), flat
    idx = torch.randperm(flat.numel(), generator=torch.Generator().manual_seed(seed))[:n]
    return idx, flat[idx]


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    paper, transformer, st_detr, set_criterion, od_map = import_reference()
    import importlib.util
    spec = importlib.util.spec_from_file_location(          # OUR generator (drop-in package), by path:
        "fod_synthetic", os.path.join(ROOT, "future-object-detection_amd", "future_od", "datasets",
                                      "synthetic.py"))      # `future_od` now names the reference
    synth = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(synth)
    make_batch = synth.make_batch

    # ---------------------------------------------------------------- G1 positional tables
    pe = paper.PositionalEncoder(no_temporal=True)
    g1 = {}
    for (h, w) in [(7, 7), (25, 42), (29, 50)]:
        t = pe.get_spatial_encoding(1, 256, h, w, "cpu")[0]            # (256,h,w)
        idx, val = subsample(t)
        g1[f"idx_{h}x{w}"], g1[f"val_{h}x{w}"] = idx, val
        g1[f"sum_{h}x{w}"] = t.double().sum()
        g1[f"abs_{h}x{w}"] = t.double().abs().sum()
    pe_t = paper.PositionalEncoder(no_temporal=False)
    offs = torch.tensor([[-1.0, -0.5, -0.25], [-2.0, -1.0, -0.5]])
    t = pe_t.get_spatio_temporal_encoding(2, 3, 64, 5, 6, "cpu", offs)
    g1["st_offsets"], g1["st_full"] = offs, t
    g1["st_noffs"] = pe_t.get_spatio_temporal_encoding(2, 3, 64, 5, 6, "cpu", None)
    save("g1_posenc", **g1)

    # ---------------------------------------------------------------- G2 query sine embedding
    ref_pts = torch.rand(16, 3, 2, generator=torch.Generator().manual_seed(1))
    save("g2_sine", pos=ref_pts, out=transformer.gen_sineembed_for_position(ref_pts, D=256),
         out64=transformer.gen_sineembed_for_position(ref_pts, D=64))

    # ---------------------------------------------------------------- G5/G7/G8/G10 full model, config 1
    # ResNet-18, 1 enc, 1 dec, 6 x 224 x 224 (BASELINE.json configs[0]) and a 2+2-layer ResNet-50 variant
    cases = {
        "g5_cfg1_r18": (Config(backbone="resnet18", enc_layers=1, dec_layers=1), 1, 6, 224, 224, 11),
        "g5_r50_2x2": (Config(backbone="resnet50", enc_layers=2, dec_layers=2), 2, 4, 96, 160, 12),
        "g5_r18_k3_noimu": (Config(backbone="resnet18", enc_layers=1, dec_layers=2, num_images=3,
                                   use_imu=False), 2, 5, 64, 96, 13),
        # one memory of all past frames + temporal encoding; temporal encoding alone
        "g11_all_at_once_temporal": (Config(backbone="resnet18", enc_layers=1, dec_layers=2, num_images=1,
                                            image_memory_mode="attend all at once", no_temporal=False),
                                     2, 4, 64, 96, 14),
        "g12_one_at_a_time_temporal": (Config(backbone="resnet18", enc_layers=1, dec_layers=2, num_images=2,
                                              no_temporal=False), 2, 4, 64, 96, 15),
        "g13_joint_encoder": (Config(backbone="resnet18", enc_layers=1, joint_layers=1, dec_layers=1, num_images=2,
                                     no_temporal=False), 2, 4, 64, 96, 16),
        # JointEncoderSequential: frame by frame with attention onto the previous output, two earlier frames and the
        # frame's IMU token (paper.py:206-234, transformer.py:463-487)
        "g14_joint_sequential": (Config(backbone="resnet18", enc_layers=1, joint_layers=2, joint_mode="sequential",
                                        joint_previmages=2, joint_prevout=True, joint_egodeep=True, dec_layers=1,
                                        num_images=2, no_temporal=False), 2, 5, 64, 96, 17),
        # the recurrent detector: decoder layers attend to the previous frame's final queries (slot states) and to
        # the frame's IMU token (transformer.py:288-307, paper.py:396-399); every past frame's pass is live
        "g15_slotstates_egodeep": (Config(backbone="resnet18", enc_layers=1, dec_layers=2, num_images=2,
                                          dec_slotstates=True, dec_egodeep=True, no_temporal=False,
                                          first_layer_special_when="first frame"), 2, 4, 64, 96, 18),
        # IMU attention with SEVERAL keys: the joint encoder's layers and the all-at-once detector attend to the IMU
        # tokens of all past frames (paper.py:196-198, 337-339)
        # SingleFrameCore (paper.py:488-528): the single-frame baseline; a three-frame clip exercises its frame loop
        # (the first frame is dead work with num_images = 2)
        "g17_single_frame_core": (Config(backbone="resnet18", enc_layers=1, dec_layers=2, num_images=2,
                                         single_frame=True), 2, 3, 64, 96, 20),
        # JointEncoderF2F (paper.py:237-277): the frames' feature maps stacked on channels through seven dilated
        # convolutions; the detector then sees one frame
        "g19_joint_f2f": (Config(backbone="resnet18", enc_layers=1, dec_layers=1, num_images=2, joint_f2f_frames=2),
                          2, 3, 96, 160, 21),
        "g16_multikey_egodeep": (Config(backbone="resnet18", enc_layers=1, joint_layers=1, joint_egodeep=True,
                                        dec_layers=2, num_images=1, image_memory_mode="attend all at once",
                                        dec_egodeep=True, no_temporal=False), 2, 4, 64, 96, 19),
        # SeparateEncoder(concat_imu=True) (paper.py:153-156): the ego-motion code is added to the frame's features
        # and no IMU token travels on
        "g20_concat_imu": (Config(backbone="resnet18", enc_layers=1, dec_layers=2, num_images=2, concat_imu=True),
                           2, 3, 64, 96, 22),
        # CDetrBackbone(dilation=True) (paper.py:95): layer4 at layer3's resolution, 3x3s of dilation 2; the 80 x 144
        # frames give layer3 a 5 x 9 map, odd both ways
        "g21_dilated_r50": (Config(backbone="resnet50", enc_layers=1, dec_layers=1, num_images=2, dilation=True),
                            2, 3, 80, 144, 23),
    }
    for name, (cfg, B, L, H, W, seed) in cases.items():
        if ONLY and name not in ONLY:
            continue
        model = build_reference(cfg, paper, transformer, st_detr)
        sd = load_weights(model, cfg, seed)
        data = make_batch(B, L, H, W, seed=seed, device="cpu", max_boxes=12)
        if not cfg.use_imu:
            for k in ("translation", "acceleration", "rotation", "rotation_rate", "speed"):
                data[k] = None
        for p in model.parameters():
            p.grad = None
        outputs, state, loss, stats, od = model(data=data, distributed=False)
        loss.backward()
        # raw core outputs again (cheap; for pred_logits / pred_boxes / aux)
        with torch.no_grad():
            kw = {}
            if cfg.use_imu:
                kw["imu"] = torch.cat([data[k] for k in model._imu_keys], dim=2)
            if not cfg.no_temporal:                       # as st_detr.py:115-118 does with encode_offset
                kw["temporal_offsets"] = data["temporal_offsets"]
            core_out, _ = model._model(data["video"], **kw)
            # dead-work equivalence (G10): feed only the last K past frames (+ future frame, unless single-frame core)
            keep = min(cfg.num_images, L) if cfg.single_frame else min(cfg.num_images, L - 1) + 1
            kw2 = {k: v[:, -keep:] for k, v in kw.items()}
            core_out_dead, _ = model._model(data["video"][:, -keep:], **kw2)
        arrays = {
            "meta": np.array([B, L, H, W, seed]),
            "pred_logits": core_out["pred_logits"], "pred_boxes": core_out["pred_boxes"],
            "dead_pred_logits": core_out_dead["pred_logits"], "dead_pred_boxes": core_out_dead["pred_boxes"],
            "loss": loss, "class_scores": outputs["class_scores"], "boxes": outputs["boxes"],
            "od_confs": od[0], "od_is_positive": od[1], "od_size_categories": od[2], "od_num_annos": od[3],
        }
        for i, aux in enumerate(core_out["aux_outputs"]):
            arrays[f"aux{i}_logits"], arrays[f"aux{i}_boxes"] = aux["pred_logits"], aux["pred_boxes"]
        for k, v in stats.items():
            arrays["stat_" + k] = v
        # gradients: norms of all, subsamples of a few (G8)
        gn = {}
        for n, p in model.named_parameters():
            if p.requires_grad:
                gn[n] = float(p.grad.double().norm()) if p.grad is not None else -1.0
        arrays["grad_names"] = np.array(list(gn.keys()))
        arrays["grad_norms"] = np.array(list(gn.values()))
        picks = [n for n in gn if n.endswith((
            "layer2.0.conv1.weight", "layer4.1.conv2.weight", "input_proj.weight",
            "layers.0.self_attn.attn.in_proj_weight", "image_attend.0.query_sine.weight",
            "image_attend.1.key_pos.weight" if cfg.num_images > 1 else "image_attend.0.key_pos.weight",
            "class_embed.weight", "ref_point_head.layers.1.weight", "query_embed.weight",
            "imu_layers.0.weight", "egodeep_attend.value.weight"))]
        named = dict(model.named_parameters())
        for n in picks:
            if named[n].grad is None:                     # e.g. the second image's attention when the detector sees one frame
                continue
            idx, val = subsample(named[n].grad, 2048)
            arrays["gidx:" + n], arrays["gval:" + n] = idx, val
        save(name, **arrays)

    # ---------------------------------------------------------------- G18 tracker baseline (paper.py:531-706)
    if not ONLY or "g18_tracker_baseline" in ONLY:
        arrays = {}
        g = torch.Generator().manual_seed(31)
        B, M, N, C = 3, 24, 20, 8                                           # rectangular: 4 current detections unmatched
        p1 = {"pred_boxes": torch.rand(B, N, 4, generator=g) * 0.6 + 0.2, "pred_logits": torch.randn(B, N, C, generator=g) * 2}
        p2 = {"pred_boxes": torch.rand(B, M, 4, generator=g) * 0.6 + 0.2, "pred_logits": torch.randn(B, M, C, generator=g) * 2}
        offs = torch.tensor([[-1.0, -0.5, 0.0], [-0.9, -0.5, 0.0], [-1.0, -0.2, 0.0]])
        arrays.update(u_b1=p1["pred_boxes"], u_l1=p1["pred_logits"], u_b2=p2["pred_boxes"], u_l2=p2["pred_logits"], u_offs=offs)
        for mode in (None, "linear", "percentual", "average"):
            for tag, o in (("none", None), ("offs", offs)):
                out = paper.TrackerFuturePredictor(mode)({k: v.clone() for k, v in p1.items()},
                                                         {k: v.clone() for k, v in p2.items()}, o)
                arrays[f"u_{mode}_{tag}_boxes"], arrays[f"u_{mode}_{tag}_logits"] = out["pred_boxes"], out["pred_logits"]
        # the whole core on three-frame clips (evaluation) and on one-frame clips (training path = single-frame core)
        cfg = Config(backbone="resnet18", enc_layers=1, dec_layers=1, num_images=1, single_frame=True, num_queries=32)
        model = build_reference(cfg, paper, transformer, st_detr)
        load_weights(model, cfg, 32)
        single = model._model
        core = paper.TrackerBaselineCore(encoder=single.encoder, detector=single.detector, pos_encoder=single.pos_encoder,
                                         tracker_future_predictor=paper.TrackerFuturePredictor("linear")).eval()
        data = make_batch(2, 3, 64, 96, seed=32, device="cpu", max_boxes=6)
        imu = torch.cat([data[k] for k in model._imu_keys], dim=2)
        offs3 = torch.tensor([[-1.0, -0.5, 0.0], [-0.8, -0.5, 0.0]])
        with torch.no_grad():
            out3, _ = core(data["video"], imu=imu, temporal_offsets=offs3)
            out1, _ = core(data["video"][:, :1], imu=imu[:, :1])
        arrays.update(meta=np.array([2, 3, 64, 96, 32]), offs3=offs3, core3_boxes=out3["pred_boxes"],
                      core3_logits=out3["pred_logits"], core1_boxes=out1["pred_boxes"], core1_logits=out1["pred_logits"])
        # the same core with the temporal positional term (PositionalEncoder(no_temporal=False)): the encoding is built
        # for the three-frame clip and sliced per detector pass (paper.py:684-700)
        cfg_t = Config(backbone="resnet18", enc_layers=1, dec_layers=1, num_images=1, single_frame=True, num_queries=32,
                       no_temporal=False)
        model_t = build_reference(cfg_t, paper, transformer, st_detr)
        load_weights(model_t, cfg_t, 32)
        st = model_t._model
        core_t = paper.TrackerBaselineCore(encoder=st.encoder, detector=st.detector, pos_encoder=st.pos_encoder,
                                           tracker_future_predictor=paper.TrackerFuturePredictor("linear")).eval()
        offs3t = torch.tensor([[0.5, 1.0, 1.5], [0.4, 0.9, 1.6]])
        with torch.no_grad():
            out3t, _ = core_t(data["video"], imu=imu, temporal_offsets=offs3t)
            out3n, _ = core_t(data["video"], imu=imu, temporal_offsets=None)
        arrays.update(offs3t=offs3t, tcore3_boxes=out3t["pred_boxes"], tcore3_logits=out3t["pred_logits"],
                      tcore3n_boxes=out3n["pred_boxes"], tcore3n_logits=out3n["pred_logits"])
        save("g18_tracker_baseline", **arrays)

    # ---------------------------------------------------------------- G3/G4 encoder & decoder stacks alone
    cfg = Config(backbone="resnet18", hidden_dim=64, nheads=4, dim_feedforward=96, enc_layers=2,
                 dec_layers=3, num_queries=20, num_images=2)
    model = build_reference(cfg, paper, transformer, st_detr)
    sd = load_weights(model, cfg, 21)
    g = torch.Generator().manual_seed(21)
    N, Bf = 35, 3
    x = torch.randn(N, Bf, 64, generator=g)
    pos = torch.randn(N, Bf, 64, generator=g)
    ego = torch.randn(1, Bf, 64, generator=g)
    with torch.no_grad():
        enc_out = model._model.separate_encoder.transformer(x, None, None, image_pos=pos, egodeep=ego)
        qpos = model._model.detector.query_embed.weight.unsqueeze(1).repeat(1, Bf, 1)
        mem2 = torch.randn(N, Bf, 64, generator=g)
        hs, ref = model._model.detector.decoder(torch.zeros_like(qpos), qpos, [x, mem2], [pos, pos], None,
                                                first_layer_special=True, egodeep=None)
        hs1, ref1 = model._model.detector.decoder(torch.zeros_like(qpos), qpos, [x], [pos], None,
                                                  first_layer_special=False, egodeep=None)
    save("g34_stacks", x=x, pos=pos, ego=ego, mem2=mem2, enc_out=enc_out, hs=hs, ref=ref, hs1=hs1, ref1=ref1)

    # ---------------------------------------------------------------- G6 matcher + G7 criterion on raw tensors
    crit_cfg = Config(dec_layers=3)
    args = st_detr.SpatioTemporalDETRArgs(num_classes=8, dec_layers=3)
    from ConditionalDETR.models.matcher import build_matcher
    crit = set_criterion.SetCriterion(8, matcher=build_matcher(args),
                                      weight_dict={}, focal_alpha=0.25,
                                      losses=["labels", "boxes", "cardinality"], matching_mode="per level")
    g6 = {}
    for ci, (B, M, nbs) in enumerate([(2, 128, [7, 23]), (3, 16, [0, 1, 40]), (1, 8, [0]), (2, 32, [32, 5])]):
        g = torch.Generator().manual_seed(100 + ci)
        logits = torch.randn(B, M, 8, generator=g) * 2 - 2
        boxes = torch.rand(B, M, 4, generator=g) * 0.5 + 0.1
        targets = []
        for nb in nbs:
            cxcy = torch.rand(nb, 2, generator=g) * 0.6 + 0.2
            wh = torch.rand(nb, 2, generator=g) * 0.3 + 0.02
            targets.append({"labels": torch.randint(0, 8, (nb,), generator=g),
                            "boxes": torch.cat([cxcy, wh], 1)})
        outputs = {"pred_logits": logits, "pred_boxes": boxes,
                   "aux_outputs": [{"pred_logits": logits.flip(1) * 0.9, "pred_boxes": boxes.flip(1)},
                                   {"pred_logits": logits * 1.1 + 0.3, "pred_boxes": boxes * 0.9 + 0.05}]}
        idx = crit.matcher({"pred_logits": logits, "pred_boxes": boxes}, targets)
        losses = crit(outputs, targets, False)
        g6[f"c{ci}_logits"], g6[f"c{ci}_boxes"] = logits, boxes
        g6[f"c{ci}_nbs"] = np.array(nbs)
        g6[f"c{ci}_tlabels"] = torch.cat([t["labels"] for t in targets])
        g6[f"c{ci}_tboxes"] = torch.cat([t["boxes"] for t in targets])
        for b, (i, j) in enumerate(idx):
            g6[f"c{ci}_i{b}"], g6[f"c{ci}_j{b}"] = i, j
        for k, v in losses.items():
            g6[f"c{ci}_L_{k}"] = v
    save("g67_criterion", **g6)

    # ---------------------------------------------------------------- G9 od_map on synthetic detections
    g9 = {}
    for ci, (B, Mp, Nmax) in enumerate([(2, 64, 10), (1, 128, 30), (3, 50, 4)]):
        g = torch.Generator().manual_seed(200 + ci)
        H, W = 448, 800
        nb = torch.randint(0, Nmax + 1, (B,), generator=g)
        xy = torch.rand(B, 256, 2, generator=g) * torch.tensor([W * 0.7, H * 0.7])
        wh = torch.rand(B, 256, 2, generator=g) * torch.tensor([W * 0.3, H * 0.3]) + 4
        aboxes = torch.cat([xy, xy + wh], 2)
        aclasses = torch.randint(0, 8, (B, 256), generator=g)
        active = (torch.arange(256)[None] < nb[:, None]).long()
        # predictions: jittered copies of annotations + noise boxes
        pxy = torch.rand(B, Mp, 2, generator=g) * torch.tensor([W * 0.7, H * 0.7])
        pwh = torch.rand(B, Mp, 2, generator=g) * torch.tensor([W * 0.3, H * 0.3]) + 4
        pboxes = torch.cat([pxy, pxy + pwh], 2)
        scores = torch.rand(B, Mp, 8, generator=g) * 0.3
        for b in range(B):
            for n in range(int(nb[b])):
                for rep in range(2):
                    m = (3 * n + rep * 17) % Mp
                    pboxes[b, m] = aboxes[b, n] + torch.randn(4, generator=g) * (3.0 + 12.0 * rep)
                    scores[b, m, aclasses[b, n]] = 0.5 + 0.5 * torch.rand((), generator=g)
        scores = torch.cat([scores, scores.max(2, keepdim=True)[0]], 2)
        out = od_map.prepare_od_map_stuffs(pboxes, scores, aboxes, aclasses, active, (H, W))
        g9[f"c{ci}_pboxes"], g9[f"c{ci}_scores"] = pboxes, scores
        g9[f"c{ci}_aboxes"], g9[f"c{ci}_aclasses"], g9[f"c{ci}_active"] = aboxes, aclasses, active
        for k, v in zip(("confs", "is_positive", "size_categories", "num_annos"), out):
            g9[f"c{ci}_{k}"] = v
        ap = np.stack([od_map._get_ap(out[0][t], out[1][t], out[2], out[3][:, :, None]).numpy()
                       for t in range(out[0].shape[0])])
        g9[f"c{ci}_ap"] = ap
    save("g9_odmap", **g9)


if __name__ == "__main__":
    main()
