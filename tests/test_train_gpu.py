"""Optimizer parity (fused clip + AdamW vs torch's clip_grad_norm_ + AdamW) and a short Trainer run
(reference future_od/trainer.py semantics: loss falls on a repeated batch, checkpoint round-trips)."""
import os
from types import SimpleNamespace

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_fused_adamw_matches_torch():
    from future_od.optim import FusedAdamW
    g = torch.Generator().manual_seed(0)
    shapes = [(300, 70), (5,), (64, 32, 3, 3), (1,), (100000,)]
    ps_a, ps_b = [], []
    for s in shapes:
        t = torch.randn(s, generator=g)
        if len(s) == 4:
            t = t.contiguous(memory_format=torch.channels_last)
        ps_a.append(torch.nn.Parameter(t.clone().to(DEV)))
        ps_b.append(torch.nn.Parameter(t.clone().to(DEV)))
    if True:
        ps_a[2].data = ps_a[2].data.contiguous(memory_format=torch.channels_last)
        ps_b[2].data = ps_b[2].data.contiguous(memory_format=torch.channels_last)
    ga = [{"params": ps_a[:2]}, {"params": ps_a[2:], "lr": 3e-4}]
    gb = [{"params": ps_b[:2]}, {"params": ps_b[2:], "lr": 3e-4}]
    a = FusedAdamW(ga, lr=1e-3, weight_decay=1e-2, max_norm=0.1)
    b = torch.optim.AdamW(gb, lr=1e-3, weight_decay=1e-2)
    for step in range(4):
        for pa, pb in zip(ps_a, ps_b):
            gr = torch.randn(pa.shape, generator=g).to(DEV)
            if pa.dim() == 4 and step % 2 == 0:
                gr = gr.contiguous(memory_format=torch.channels_last)   # both layouts of the same values
            pa.grad, pb.grad = gr.clone(), gr.clone()
        torch.nn.utils.clip_grad_norm_(ps_b, 0.1)
        a.step(); b.step()
    for pa, pb in zip(ps_a, ps_b):
        torch.testing.assert_close(pa, pb, rtol=2e-6, atol=2e-7)
    sa, sb = a.state_dict(), b.state_dict()
    assert set(sa["state"][0].keys()) >= {"step", "exp_avg", "exp_avg_sq"} and len(sa["param_groups"]) == 2
    torch.testing.assert_close(sa["state"][3]["exp_avg_sq"], sb["state"][3]["exp_avg_sq"], rtol=1e-5, atol=1e-9)


class _Loader(list):
    batch_size = 2


def test_trainer_short_run(tmp_path):
    from future_od.datasets.synthetic import make_batch
    from future_od.models.st_detr import SpatioTemporalDETRArgs
    from future_od.trainer import Trainer
    from runs._helper import get_lr_func, setup_optimizer
    from runs._model import build_model
    torch.manual_seed(0)
    args = SimpleNamespace(device=DEV, distributed=False, compute_dtype="bf16", backbone="resnet18")
    detr = SpatioTemporalDETRArgs(num_classes=8, num_queries=32, lr_backbone=1e-4, enc_layers=1, dec_layers=2,
                                  pretrained_backbone=False)
    model = build_model(args, detr)
    sched, opt = setup_optimizer(detr, model, get_lr_func(4))
    batch = make_batch(2, 3, 96, 128, seed=1, max_boxes=5)
    loader = _Loader([batch] * 6)
    tr = Trainer(model, opt, sched, loader, {"val": _Loader([batch])}, str(tmp_path), str(tmp_path), "t", DEV,
                 print_interval=3, visualization_epochs=[], visualization_iterations=[], category_dict={},
                 checkpoint_epochs=True, is_master=True, max_norm=detr.max_norm)
    before = {k: v.clone() for k, v in model.state_dict().items()}
    tr.train(2)
    # the training epochs ran as replays of ONE captured step (train mode, dropout 0.1 active), one update per batch:
    # the capture's warm-up steps were rolled back
    assert tr._graphed not in (None, False) and tr._graphed.replays == 12 and len(tr._graphed._graphs) == 1
    assert opt._step_no == 12 and tr._training_iterations == 12
    # ... and the two evaluation passes as replays of one captured forward
    assert tr._graphed_eval not in (None, False) and tr._graphed_eval.replays == 2 and len(tr._graphed_eval._graphs) == 1
    hist = tr._stats["train labels loss"].history
    assert len(hist) == 2 and all(h == h for h in hist)            # finite
    total = [sum(tr._stats[f"train {k} loss"].history[e] for k in ("labels", "box_l1", "box_giou")) for e in range(2)]
    assert total[1] < total[0], total                              # learning on a repeated batch
    changed = sum(int(not torch.equal(before[k], v)) for k, v in model.state_dict().items())
    assert changed > 100
    frozen = "_model.separate_encoder.backbone.body.layer1.0.conv1.weight"
    assert torch.equal(before[frozen], model.state_dict()[frozen])
    assert os.path.isfile(tmp_path / "t.pth.tar") and os.path.isfile(tmp_path / "t_final.pth.tar")
    model2 = build_model(args, detr)
    sched2, opt2 = setup_optimizer(detr, model2, get_lr_func(4))
    tr2 = Trainer(model2, opt2, sched2, loader, {"val": _Loader([batch])}, str(tmp_path), str(tmp_path), "t", DEV,
                  print_interval=3, visualization_epochs=[], visualization_iterations=[], category_dict={},
                  is_master=True)
    tr2.load_checkpoint()
    assert tr2._epoch == 2
    for k, v in model.state_dict().items():
        assert torch.equal(v, model2.state_dict()[k]), k
    assert hasattr(tr, "last_ap") and tr.last_ap["all"].shape[0] == 10


def test_device_prefetcher_stages_next_batch():
    """Batches staged through pinned memory on a side stream arrive intact, keep their host annotation copies, and
    the model consumes them (uint8 and float clips alike)."""
    import torch
    from future_od.datasets.synthetic import make_batch
    from future_od.utils.prefetch import DevicePrefetcher
    host_batches = []
    for s in range(3):
        b = make_batch(1, 2, 32, 48, seed=s, max_boxes=4)
        b.pop("_host_annotations")
        if s == 1:
            b["video"] = torch.randint(0, 256, b["video"].shape, dtype=torch.uint8)
        host_batches.append(b)
    got = list(DevicePrefetcher(host_batches, "cuda:0"))
    torch.cuda.synchronize()
    assert len(got) == 3
    for src, dst in zip(host_batches, got):
        assert dst["video"].is_cuda and dst["video"].dtype == src["video"].dtype
        assert torch.equal(dst["video"].cpu(), src["video"]) and torch.equal(dst["boxes"].cpu(), src["boxes"])
        assert dst["_host_annotations"]["classes"] is src["classes"]


def test_nusc_500ms_run_script_two_stages(tmp_path, capsys):
    """BASELINE.json configs[3]: the reference's 500 ms NuScenes experiment (IMU token fusion, three-frame clips, two
    resolution stages) end to end on synthetic NuScenes-shaped batches, shrunk to a smoke run."""
    import importlib
    run = importlib.import_module("runs.nusc_spatiotemporal_imu_500ms")
    tr = run.main(["--epochs", "2", "--steps_per_epoch", "2", "--val_steps", "1", "--out", str(tmp_path),
                   "--stage_sizes", "64x96:4,96x128:2", "--device", DEV])
    out = capsys.readouterr().out
    assert "Starting first training stage" in out and "Starting second training stage" in out
    assert tr._epoch == 2 and tr._training_iterations == 4
    hist = tr._stats["train labels loss"].history
    assert len(hist) == 2 and all(h == h for h in hist)                      # both stages ran, finite losses
    assert tr._train_loader.dataset.size == (96, 128) and tr._train_loader.batch_size == 2
    assert os.path.isfile(tmp_path / "nusc_spatiotemporal_imu_500ms_final.pth.tar")
    assert hasattr(tr, "last_ap")                                             # AP50 bookkeeping ran on the val pass
