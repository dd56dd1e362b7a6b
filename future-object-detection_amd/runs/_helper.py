"""Optimizer / schedule / argument helpers with the reference's names (reference runs/_helper.py)."""
import argparse
import os

import numpy as np
import torch
import torch.optim as optim

from future_od.optim import FusedAdamW


def get_lr_func(epochs):
    """10 % linear warm-up, x0.5 after 60 %, x0.1 after 90 % of the epochs (reference runs/_helper.py:69-81)."""
    warmup, drop_1, drop_2 = int(0.1 * epochs), int(0.6 * epochs), int(0.9 * epochs)

    def f(e):
        if e < warmup:
            return (e + 1) / (1 + warmup)
        return 1 if e <= drop_1 else (0.5 if e <= drop_2 else 0.1)

    return f


def setup_optimizer(detr_args, model, lr_func):
    """AdamW with a separate learning rate for parameters whose name contains "backbone" (which includes
    input_proj), reference runs/_helper.py:84-107; the update itself is the fused HIP kernel pair."""
    m = model.module if isinstance(model, torch.nn.parallel.DistributedDataParallel) else model
    named = [(n, p) for n, p in m.named_parameters() if p.requires_grad]
    groups = [{"params": [p for n, p in named if "backbone" not in n]},
              {"params": [p for n, p in named if "backbone" in n], "lr": detr_args.lr_backbone}]
    optimizer = FusedAdamW(groups, lr=detr_args.lr, weight_decay=detr_args.weight_decay,
                           max_norm=detr_args.max_norm)
    return optim.lr_scheduler.LambdaLR(optimizer, lr_func), optimizer


def add_pytorch_args(parser):
    parser.add_argument("-d", "--device", dest="device", type=str, default="cuda:0")
    parser.add_argument("--distributed", action="store_true", default=False)
    parser.add_argument("--local_rank", default=0, type=int)
    parser.add_argument("--world_size", default=1, type=int)
    parser.add_argument("--dist_url", default="env://", type=str)


def build_base_parser():
    parser = argparse.ArgumentParser(description="Experiment runfile, you run experiments from this file")
    for flag in ("--restart", "--debug", "--disable_wandb", "--no_checkpoints", "--short_train", "--night",
                 "--load-only-net"):
        parser.add_argument(flag, action="store_true", default=False)
    parser.add_argument("--wandb_resume_id", default=None)
    parser.add_argument("--checkpoint", default=None, help="Override checkpoint to be loaded")
    parser.add_argument("--compute_dtype", default="bf16", choices=["bf16", "fp32"])
    add_pytorch_args(parser)
    return parser


def get_trainer(args, config, detr_args, lr_sched, model, optimizer, train_loader, val_loaders):
    """The Trainer as the reference's run scripts build it (reference runs/_helper.py:15-66)."""
    from future_od.trainer import Trainer
    from future_od.utils.wandb import WandBConfig
    from runs._loader import CATEGORY_DICT
    return Trainer(
        model=model, optimizer=optimizer, lr_sched=lr_sched, train_loader=train_loader, val_loaders=val_loaders,
        checkpoint_path=config["checkpoint_path"],
        visualization_path=os.path.join(config["visualization_path"], args.experiment_idf),
        save_name=args.experiment_idf, device=args.device, checkpoint_epochs=not args.no_checkpoints,
        print_interval=25, visualization_epochs=set(int(i) for i in np.linspace(1, args.epochs, 10)),
        visualization_iterations=[0], category_dict=CATEGORY_DICT, distributed=args.distributed,
        is_master=(args.world_rank == 0),
        wandb_config=WandBConfig(enabled=False), max_norm=detr_args.max_norm)
