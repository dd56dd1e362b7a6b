"""The reference's 500 ms NuScenes experiment (reference runs/nusc_spatiotemporal_imu_500ms.py) on synthetic
NuScenes-shaped batches: clips of three frames at offsets (-1.0 s, -0.5 s, 0), IMU token fusion, two stages --
60 % of the epochs at 448x800 with global batch 32, the rest at 896x1600 with global batch 16 -- 128 queries, 8 classes,
lr 1e-4 / backbone 1e-4.  BASELINE.json configs[3].

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \\
        future-object-detection_amd/runs/nusc_spatiotemporal_imu_500ms.py --distributed --epochs 160
    (one GPU: drop the launcher and --distributed; --steps_per_epoch / --stage_sizes shrink it for a smoke run)
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from future_od.models.st_detr import SpatioTemporalDETRArgs  # noqa: E402
from future_od.utils.distributed import init_distributed_and_device_  # noqa: E402
from runs._helper import build_base_parser, get_lr_func, get_trainer, setup_optimizer  # noqa: E402
from runs._loader import CATEGORY_DICT, get_nusc_loaders  # noqa: E402
from runs._model import build_model  # noqa: E402

OFFSETS = [-1.0, -0.5, 0]
STAGES = (((448, 800), 32), ((896, 1600), 16))          # (size, global batch), reference :19-25,34-40


def train(model, args, detr_args, config):
    """Stage k runs until `until[k]` epochs have been done in total (60 % / 100 %): the learning-rate schedule spans
    both stages, the loaders are swapped in between (reference :17-41)."""
    schedule, optimizer = setup_optimizer(detr_args, model, get_lr_func(args.epochs))
    stages = getattr(args, "stages", STAGES)
    until = (int(args.epochs * 0.60), args.epochs)
    trainer = None
    for k, ((size, global_batch), last_epoch) in enumerate(zip(stages, until)):
        loaders = get_nusc_loaders(size, offsets=OFFSETS, config=config, args=args, train_batch_size=global_batch)
        if trainer is None:
            trainer = get_trainer(args, config, detr_args, schedule, model, optimizer, *loaders)
        else:
            trainer._train_loader, trainer._val_loaders = loaders
        print(f"Starting {('first', 'second')[k]} training stage: {size[0]}x{size[1]}, global batch {global_batch}, "
              f"epochs up to {last_epoch}")
        trainer.train(last_epoch)
    return trainer


def main(argv=None):
    print(f"{os.path.basename(__file__)}: pytorch {torch.__version__}, hip {torch.version.hip}")
    parser = build_base_parser()
    parser.add_argument("--epochs", default=160, type=int, help="Number of training epochs")
    parser.add_argument("--steps_per_epoch", default=8, type=int, help="synthetic batches per epoch")
    parser.add_argument("--val_steps", default=2, type=int)
    parser.add_argument("--out", default="/tmp/fod_runs", help="checkpoint / visualisation directory")
    parser.add_argument("--stage_sizes", default=None,
                        help="override the two stages, 'H1xW1:B1,H2xW2:B2' (global batches), e.g. for a smoke run")
    args = parser.parse_args(argv)
    if args.stage_sizes:
        args.stages = tuple((tuple(int(v) for v in sz.split("x")), int(b))
                            for sz, b in (item.split(":") for item in args.stage_sizes.split(",")))
        assert len(args.stages) == 2
    args.experiment_idf = os.path.splitext(os.path.basename(__file__))[0]
    detr_args = SpatioTemporalDETRArgs(num_classes=len(CATEGORY_DICT), num_queries=128, lr_backbone=1e-4,
                                       pretrained_backbone=False)
    init_distributed_and_device_(args)
    model = build_model(args, detr_args)
    os.makedirs(args.out, exist_ok=True)
    config = {"checkpoint_path": args.out, "visualization_path": args.out}
    return train(model, args, detr_args, config)


if __name__ == "__main__":
    main()
