from dataclasses import dataclass, field


@dataclass
class WandBConfig:
    enabled: bool = False
    project: str = "future-od"
    entity: str = None
    name: str = None
    notes: str = None
    resume_id: str = None
    hyperparams: dict = field(default_factory=dict)
    watch_model: bool = False
    num_images: int = 4
