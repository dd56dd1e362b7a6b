"""Settings object for the optional Weights & Biases logging of the Trainer.

Logging is off unless `enabled` is set AND the `wandb` package is importable (it is not in the build / GPU images; the
Trainer then only prints).  Field names are the ones the reference's run scripts pass (runs/_helper.py:50-63)."""
import dataclasses
from typing import Optional


@dataclasses.dataclass
class WandBConfig:
    enabled: bool = False                       # master switch; False = never import or touch wandb
    name: Optional[str] = None                  # run name (the experiment id of the run script)
    notes: Optional[str] = None
    num_images: int = 4                         # visualisations to upload per epoch
    hyperparams: dict = dataclasses.field(default_factory=dict)
    project: str = "future-od"
    entity: Optional[str] = None
    resume_id: Optional[str] = None             # continue an existing run
    watch_model: bool = False                   # wandb.watch gradients / parameters
