"""The NuScenes side of the drop-in surface (reference future_od/datasets/nu_scenes.py): the category table the run
scripts size the model with (`len(nu_scenes.CATEGORY_DICT)`, reference runs/nusc_spatiotemporal_imu_500ms.py:53) and the
batch schema (reference :334-351).  The dataset READER itself (nuscenes-devkit, the data) is out of scope:
`NuScenesDataset` here is the synthetic stand-in that yields NuScenes-shaped clips -- same keys, shapes and dtypes."""
from future_od.datasets.synthetic import make_batch

# reference future_od/datasets/nu_scenes.py:29-39
CATEGORY_DICT = {
    0: "Vehicle",
    1: "Truck",
    2: "Trailer",
    3: "Pedestrian",
    4: "Bus",
    5: "Motorcyclist",
    6: "Bicyclist",
    7: "ConstructionVehicle",
}
IGNORE_CATEGORY = len(CATEGORY_DICT)
ORIGINAL_IMSIZE = (900, 1600)


class NuScenesDataset:
    """Synthetic NuScenes-shaped clips: item i is one sample of the reference's schema (video [L,3,H,W] + IMU rows +
    dense annotations padded to 256 slots); deterministic in (seed, i).  Constructor keywords follow the reference's
    (`root_path`, `split`, `frame_offsets`, ...); the ones that only make sense with real data are accepted and ignored."""

    def __init__(self, root_path=None, split="train", night=False, front_camera_only=True, joint_transform=None,
                 frame_offsets=(-1.0, -0.5, 0), annotated_frame_idx_override=None, filter_offsets=None, size=(448, 800),
                 length=64, seed=1234, max_boxes=40):
        self.size, self.offsets, self.length, self.seed, self.max_boxes = tuple(size), list(frame_offsets), int(length), seed, max_boxes

    def __len__(self):
        return self.length

    def __getitem__(self, i):
        b = make_batch(1, len(self.offsets), self.size[0], self.size[1], seed=self.seed + 7919 * int(i), max_boxes=self.max_boxes)
        b.pop("_host_annotations", None)
        return {k: (v[0] if hasattr(v, "shape") else v) for k, v in b.items()}
