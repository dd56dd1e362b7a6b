"""The NuImages side of the drop-in surface (reference future_od/datasets/nu_images.py:15-29): constants the run scripts
use (`nu_images.ANNOTATED_FRAME + offset`, `CATEGORY_DICT`); the reader itself is out of scope (no data, no devkit),
the synthetic NuScenes-shaped dataset stands in."""
from future_od.datasets.nu_scenes import CATEGORY_DICT, IGNORE_CATEGORY, NuScenesDataset  # noqa: F401

ORIGINAL_IMSIZE = (900, 1600)
ANNOTATED_FRAME = 6        # 6 frames before (0-5), 6 after (7-12)


class NuImagesDataset(NuScenesDataset):
    def __init__(self, root_path=None, split="train", night=False, front_camera_only=True, joint_transform=None,
                 frames=(4, 5, 6), max_frame_random_offset=0, annotated_frame_idx_override=None, **kw):
        super().__init__(root_path=root_path, split=split, frame_offsets=[f - ANNOTATED_FRAME for f in frames], **kw)
