"""The CPU oracle against the fixtures captured from the reference's own files
(tests/golden/make_golden.py).  Tolerances: the oracle and the reference both run fp32 torch
ops on CPU but in different groupings (functional vs nn.Module, batch layouts), so values
agree to fp32 rounding accumulated through the network: 2e-5 abs / 1e-4 rel on outputs."""
import numpy as np
import pytest
import torch

from future_od.datasets.synthetic import make_batch
from oracle import criterion as ocrit
from oracle import od_map as ood
from oracle import stdetr as O
from oracle.stdetr import Config

torch.set_num_threads(8)


def close(a, b, atol=2e-5, rtol=1e-4):
    a = a.detach().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    np.testing.assert_allclose(a, b, atol=atol, rtol=rtol)


@pytest.mark.parametrize("hw", [(7, 7), (25, 42), (29, 50)])
def test_g1_spatial_table(golden, hw):
    g = golden("g1_posenc")
    h, w = hw
    t = O.spatial_pos_table(h, w, 256)
    close(t.reshape(-1)[g[f"idx_{h}x{w}"]], g[f"val_{h}x{w}"], atol=1e-6)
    assert abs(float(t.double().sum()) - float(g[f"sum_{h}x{w}"])) < 1e-2
    assert abs(float(t.double().abs().sum()) - float(g[f"abs_{h}x{w}"])) < 1e-2


def test_g1_temporal_table(golden):
    g = golden("g1_posenc")
    sp = O.spatial_pos_table(5, 6, 64)[None, None]
    offs = torch.from_numpy(g["st_offsets"])
    close(sp + O.temporal_pos_table(2, 3, 5, 6, 64, offs), g["st_full"], atol=1e-6)
    close(sp + O.temporal_pos_table(2, 3, 5, 6, 64, None), g["st_noffs"], atol=1e-6)


def test_g2_query_sine(golden):
    g = golden("g2_sine")
    pos = torch.from_numpy(g["pos"])
    close(O.query_sine_embed(pos, 256), g["out"], atol=1e-6)
    close(O.query_sine_embed(pos, 64), g["out64"], atol=1e-6)


def test_g34_encoder_decoder_stacks(golden):
    g = golden("g34_stacks")
    cfg = Config(backbone="resnet18", hidden_dim=64, nheads=4, dim_feedforward=96, enc_layers=2,
                 dec_layers=3, num_queries=20, num_images=2)
    sd = O.make_state_dict(cfg, 21)
    x, pos, ego, mem2 = (torch.from_numpy(g[k]) for k in ("x", "pos", "ego", "mem2"))
    with torch.no_grad():
        y = x
        for i in range(cfg.enc_layers):
            y = O.encoder_layer(sd, cfg, i, y, pos, ego)
        close(y, g["enc_out"])
        qpos = sd[O.P_DET + "query_embed.weight"].unsqueeze(1).repeat(1, x.shape[1], 1)
        hs, ref = O.decoder_forward(sd, cfg, torch.zeros_like(qpos), qpos, [x, mem2], [pos, pos], True)
        close(hs, g["hs"]); close(ref, g["ref"])
        hs1, ref1 = O.decoder_forward(sd, cfg, torch.zeros_like(qpos), qpos, [x], [pos], False)
        close(hs1, g["hs1"]); close(ref1, g["ref1"])


CASES = {
    "g5_cfg1_r18": Config(backbone="resnet18", enc_layers=1, dec_layers=1),
    "g5_r50_2x2": Config(backbone="resnet50", enc_layers=2, dec_layers=2),
    "g5_r18_k3_noimu": Config(backbone="resnet18", enc_layers=1, dec_layers=2, num_images=3, use_imu=False),
    # SURVEY 8(f)-2 variants (paper.py:66-73 temporal encoding, :334-339 one memory of all past frames)
    "g11_all_at_once_temporal": Config(backbone="resnet18", enc_layers=1, dec_layers=2, num_images=1,
                                       image_memory_mode="attend all at once", no_temporal=False),
    "g12_one_at_a_time_temporal": Config(backbone="resnet18", enc_layers=1, dec_layers=2, num_images=2,
                                         no_temporal=False),
    "g19_joint_f2f": Config(backbone="resnet18", enc_layers=1, dec_layers=1, num_images=2, joint_f2f_frames=2),
    "g13_joint_encoder": Config(backbone="resnet18", enc_layers=1, joint_layers=1, dec_layers=1, num_images=2,
                                no_temporal=False),
    "g14_joint_sequential": Config(backbone="resnet18", enc_layers=1, joint_layers=2, joint_mode="sequential",
                                   joint_previmages=2, joint_prevout=True, joint_egodeep=True, dec_layers=1,
                                   num_images=2, no_temporal=False),
    "g15_slotstates_egodeep": Config(backbone="resnet18", enc_layers=1, dec_layers=2, num_images=2, dec_slotstates=True,
                                     dec_egodeep=True, no_temporal=False, first_layer_special_when="first frame"),
    "g16_multikey_egodeep": Config(backbone="resnet18", enc_layers=1, joint_layers=1, joint_egodeep=True, dec_layers=2,
                                   num_images=1, image_memory_mode="attend all at once", dec_egodeep=True,
                                   no_temporal=False),
    "g17_single_frame_core": Config(backbone="resnet18", enc_layers=1, dec_layers=2, num_images=2, single_frame=True),
    "g20_concat_imu": Config(backbone="resnet18", enc_layers=1, dec_layers=2, num_images=2, concat_imu=True),
    "g21_dilated_r50": Config(backbone="resnet50", enc_layers=1, dec_layers=1, num_images=2, dilation=True),
}


@pytest.mark.parametrize("name", list(CASES))
def test_g5_full_model_loss_and_grads(golden, name):
    g = golden(name)
    cfg = CASES[name]
    B, L, H, W, seed = (int(v) for v in g["meta"])
    sd = O.make_state_dict(cfg, seed)
    spec = O.param_spec(cfg)
    for k, (_, kind) in spec.items():
        if kind == "param":
            sd[k].requires_grad_(True)
    by_ref_name = O.rename_for_core(sd, cfg)           # the fixture names parameters as the reference's core does
    data = make_batch(B, L, H, W, seed=seed, max_boxes=12)
    imu = O.imu_from_data(data) if cfg.use_imu else None
    offs = None if cfg.no_temporal else data["temporal_offsets"]          # st_detr.py passes them with encode_offset
    out = O.core_forward(sd, cfg, data["video"], imu, offs)
    close(out["pred_logits"], g["pred_logits"])
    close(out["pred_boxes"], g["pred_boxes"])
    for i, aux in enumerate(out["aux_outputs"]):
        close(aux["pred_logits"], g[f"aux{i}_logits"])
        close(aux["pred_boxes"], g[f"aux{i}_boxes"])
    loss, stats, _ = ocrit.total_loss(cfg, out, data)
    close(loss, g["loss"], atol=1e-4, rtol=1e-5)
    for k, v in stats.items():
        close(v, g["stat_" + k], atol=1e-4, rtol=1e-5)
    loss.backward()
    names, norms = list(g["grad_names"]), g["grad_norms"]
    for n, ref_norm in zip(names, norms):
        got = by_ref_name[n].grad
        if ref_norm < 0:                       # reference left .grad = None
            assert got is None or float(got.abs().max()) == 0.0, n
            continue
        assert got is not None, n
        assert abs(float(got.double().norm()) - ref_norm) <= 1e-4 * max(ref_norm, 1e-3) + 1e-6, n
    for k in g.files:
        if k.startswith("gidx:"):
            n = k[5:]
            close(by_ref_name[n].grad.reshape(-1)[g[k]], g["gval:" + n], atol=1e-5, rtol=2e-4)
    # G10: dead-work equivalence -- both in the reference (fixture) and in the oracle.  With one memory of all past
    # frames, a joint encoder or slot states nothing is dead (the fixture's truncated run then differs, as it must).
    if cfg.image_memory_mode == "attend all at once" or cfg.joint_layers or cfg.dec_slotstates:
        assert float(np.abs(g["dead_pred_logits"] - g["pred_logits"]).max()) > 1e-3
        return
    close(g["dead_pred_logits"], g["pred_logits"])   # different frame count => different conv blocking => fp32 rounding only
    with torch.no_grad():
        out2 = O.core_forward(sd, cfg, data["video"], imu, offs, skip_dead=True)
    close(out2["pred_logits"], g["pred_logits"]); close(out2["pred_boxes"], g["pred_boxes"])


def test_g67_matcher_and_criterion(golden):
    g = golden("g67_criterion")
    cfg = Config(dec_layers=3)
    for ci in range(4):
        logits, boxes = torch.from_numpy(g[f"c{ci}_logits"]), torch.from_numpy(g[f"c{ci}_boxes"])
        nbs = [int(v) for v in g[f"c{ci}_nbs"]]
        tl = torch.from_numpy(g[f"c{ci}_tlabels"]).split(nbs)
        tb = torch.from_numpy(g[f"c{ci}_tboxes"]).split(nbs)
        targets = [{"labels": a, "boxes": b} for a, b in zip(tl, tb)]
        idx = ocrit.hungarian_match(cfg, logits, boxes, targets)
        for b, (i, j) in enumerate(idx):
            assert np.array_equal(i.numpy(), g[f"c{ci}_i{b}"])      # bit-exact indices
            assert np.array_equal(j.numpy(), g[f"c{ci}_j{b}"])
            assert np.all(np.diff(i.numpy()) > 0) or len(i) <= 1    # rows ascending like scipy
        outputs = {"pred_logits": logits, "pred_boxes": boxes,
                   "aux_outputs": [{"pred_logits": logits.flip(1) * 0.9, "pred_boxes": boxes.flip(1)},
                                   {"pred_logits": logits * 1.1 + 0.3, "pred_boxes": boxes * 0.9 + 0.05}]}
        losses = ocrit.set_criterion(cfg, outputs, targets)
        keys = [k[len(f"c{ci}_L_"):] for k in g.files if k.startswith(f"c{ci}_L_")]
        assert set(keys) == set(losses.keys())
        for k in keys:
            close(losses[k], g[f"c{ci}_L_{k}"], atol=1e-6, rtol=1e-5)


def test_g9_od_map(golden):
    g = golden("g9_odmap")
    for ci in range(3):
        out = ood.prepare_od_map_stuffs(g[f"c{ci}_pboxes"], g[f"c{ci}_scores"], g[f"c{ci}_aboxes"],
                                        g[f"c{ci}_aclasses"], g[f"c{ci}_active"], (448, 800))
        np.testing.assert_allclose(out[0], g[f"c{ci}_confs"], atol=0, rtol=0)
        assert np.array_equal(out[1], g[f"c{ci}_is_positive"])
        assert np.array_equal(out[2], g[f"c{ci}_size_categories"])
        assert np.array_equal(out[3], g[f"c{ci}_num_annos"])
        for t in range(out[0].shape[0]):
            ap = ood.average_precision(out[0][t], out[1][t], out[2], out[3][:, :, None])
            np.testing.assert_allclose(ap, g[f"c{ci}_ap"][t], atol=1e-6, rtol=1e-5, equal_nan=True)


def test_g18_tracker_baseline(golden):
    """TrackerFuturePredictor (all four size-extrapolation modes, with and without temporal offsets, rectangular
    assignment) and TrackerBaselineCore on three-frame and one-frame clips, reference paper.py:531-706."""
    g = golden("g18_tracker_baseline")
    t = lambda k: torch.from_numpy(g[k])
    p1 = {"pred_boxes": t("u_b1"), "pred_logits": t("u_l1")}
    p2 = {"pred_boxes": t("u_b2"), "pred_logits": t("u_l2")}
    for mode in (None, "linear", "percentual", "average"):
        for tag, offs in (("none", None), ("offs", t("u_offs"))):
            out = O.tracker_future_predictor(p1, p2, offs, mode)
            close(out["pred_boxes"], g[f"u_{mode}_{tag}_boxes"])
            close(out["pred_logits"], g[f"u_{mode}_{tag}_logits"])
    cfg = Config(backbone="resnet18", enc_layers=1, dec_layers=1, num_images=1, single_frame=True, num_queries=32)
    B, L, H, W, seed = (int(v) for v in g["meta"])
    sd = O.make_state_dict(cfg, seed)
    data = make_batch(B, L, H, W, seed=seed, max_boxes=6)
    imu = O.imu_from_data(data)
    with torch.no_grad():
        out3 = O.tracker_core_forward(sd, cfg, data["video"], imu, t("offs3"), "linear")
        out1 = O.tracker_core_forward(sd, cfg, data["video"][:, :1], imu[:, :1])
    close(out3["pred_boxes"], g["core3_boxes"]); close(out3["pred_logits"], g["core3_logits"])
    close(out1["pred_boxes"], g["core1_boxes"]); close(out1["pred_logits"], g["core1_logits"])
    # with the temporal positional term: built over the three-frame clip, sliced per detector pass (paper.py:684-700)
    cfg_t = Config(backbone="resnet18", enc_layers=1, dec_layers=1, num_images=1, single_frame=True, num_queries=32,
                   no_temporal=False)
    with torch.no_grad():
        out3t = O.tracker_core_forward(sd, cfg_t, data["video"], imu, t("offs3t"), "linear")
        out3n = O.tracker_core_forward(sd, cfg_t, data["video"], imu, None, "linear")
    close(out3t["pred_boxes"], g["tcore3_boxes"]); close(out3t["pred_logits"], g["tcore3_logits"])
    close(out3n["pred_boxes"], g["tcore3n_boxes"]); close(out3n["pred_logits"], g["tcore3n_logits"])
