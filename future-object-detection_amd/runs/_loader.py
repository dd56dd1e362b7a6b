"""Loaders with the reference's call shape (reference runs/_loader.py:10-95: `get_nusc_loaders(img_size, offsets, args,
config, train_batch_size, random_aug, val_annotated_frame_override, filter_offsets)` and `get_nuim_loaders(...)`),
drawing SYNTHETIC NuScenes-shaped batches: there is no dataset and no network on the
build / GPU boxes (BASELINE.json configs[3]: "random-init weights, synthetic NuScenes-shaped batches").

Sharding follows the reference (`runs/_loader.py:110-112`): the per-GPU batch is the global batch divided by the world
size, and every rank draws different samples (seed = base + epoch-independent sample id, DistributedSampler-like).
A clip has one frame per offset (`offsets=[-1.0, -0.5, 0]` -> L = 3: two past frames and the annotated one)."""
import torch

from future_od.datasets.synthetic import make_batch

from future_od.datasets.nu_scenes import CATEGORY_DICT      # noqa: F401  (reference nu_scenes.py:29-38)


class SyntheticNuScenes:
    """Stands where the reference's NuScenesDataset stands (`loader.dataset`)."""

    def __init__(self, size, offsets, length):
        self.size, self.offsets, self.length = tuple(size), list(offsets), int(length)

    def __len__(self):
        return self.length


class SyntheticLoader:
    """Iterable of `steps` batches of `batch_size` clips; batch i of rank r is seeded by (seed, i, r), so a loader
    yields the same data every epoch (like a dataset without augmentation) and ranks never share samples."""

    def __init__(self, size, offsets, batch_size, steps, rank=0, world=1, seed=1234, uint8=False, max_boxes=40):
        self.dataset = SyntheticNuScenes(size, offsets, steps * batch_size * world)
        self.batch_size, self.steps = int(batch_size), int(steps)
        self.rank, self.world, self.seed = rank, world, seed
        self.uint8, self.max_boxes = uint8, max_boxes

    def __len__(self):
        return self.steps

    def __iter__(self):
        H, W = self.dataset.size
        L = len(self.dataset.offsets)
        for i in range(self.steps):
            b = make_batch(self.batch_size, L, H, W, seed=self.seed + 7919 * (i * self.world + self.rank),
                           max_boxes=self.max_boxes, video_dtype=torch.uint8 if self.uint8 else torch.float32)
            b["temporal_offsets"] = torch.tensor(self.dataset.offsets, dtype=torch.float32).repeat(self.batch_size, 1)
            yield b


def get_nusc_loaders(img_size, offsets, args, config, train_batch_size, random_aug=None,
                     val_annotated_frame_override=None, filter_offsets=None, val_batch_size=None, steps_per_epoch=None,
                     val_steps=None):
    """-> (train_loader, {"val": val_loader}); `train_batch_size` is GLOBAL (reference :110-112).  `offsets` may be a
    dict {"train": ..., "val": ...} (reference :64-68); the augmentation / filter arguments exist for call
    compatibility and are ignored by the synthetic source."""
    size = img_size
    if isinstance(offsets, dict):
        assert "train" in offsets and "val" in offsets
        offsets, val_offsets = offsets["train"], offsets["val"]
    else:
        val_offsets = offsets
    world = getattr(args, "world_size", 1) if getattr(args, "distributed", False) else 1
    rank = getattr(args, "world_rank", 0)
    assert train_batch_size % world == 0, "global batch must divide over the ranks"
    per_gpu = train_batch_size // world
    steps = steps_per_epoch or getattr(args, "steps_per_epoch", 8)
    vsteps = val_steps or getattr(args, "val_steps", 2)
    train = SyntheticLoader(size, offsets, per_gpu, steps, rank, world, seed=1234)
    val = SyntheticLoader(size, val_offsets, val_batch_size or per_gpu, vsteps, rank, world, seed=99991)
    return train, {"val": val}


def get_nuim_loaders(img_size, offsets, args, config, train_batch_size, random_aug=None,
                     val_annotated_frame_override=None, **kw):
    """Reference runs/_loader.py:10-50: NuImages clips are addressed by FRAME INDEX offsets around the annotated frame
    (`nu_images.ANNOTATED_FRAME + offset`); only their count matters to the synthetic source."""
    return get_nusc_loaders(img_size, offsets, args, config, train_batch_size, random_aug,
                            val_annotated_frame_override, **kw)
