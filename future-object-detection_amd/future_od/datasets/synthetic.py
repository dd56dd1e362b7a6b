"""Synthetic NuScenes-shaped batches.

There is no dataset (and no network) on the build or GPU boxes, so every test and the
benchmark run on batches drawn here.  The dict has the schema the reference's dataset
emits (reference future_od/datasets/nu_scenes.py:334-351, dense 256-slot targets as in
future_od/datasets/utils.py:19-38); value distributions are those fixed in SURVEY.md 8(d).
"""
import torch

MAX_NUM_OBJECTS = 256


def _quat_mul(a, b):
    aw, ax, ay, az = a.unbind(-1)
    bw, bx, by, bz = b.unbind(-1)
    return torch.stack([
        aw * bw - ax * bx - ay * by - az * bz,
        aw * bx + ax * bw + ay * bz - az * by,
        aw * by - ax * bz + ay * bw + az * bx,
        aw * bz + ax * by - ay * bx + az * bw,
    ], dim=-1)


def make_batch(B, L, H, W, seed=1234, device="cpu", num_classes=8, max_boxes=40, min_boxes=1,
               video_dtype=torch.float32):
    """One batch of B clips of L frames.  Generated on CPU (seeded), then moved to `device`."""
    g = torch.Generator().manual_seed(seed)
    video = torch.randn(B, L, 3, H, W, generator=g, dtype=torch.float32).to(video_dtype)
    translation = (0.5 * torch.randn(B, L, 3, generator=g)).cumsum(1)
    translation = translation - translation[:, :1]
    q = torch.randn(B, L, 4, generator=g)
    q = q / q.norm(dim=-1, keepdim=True)
    q0_inv = q[:, :1] * torch.tensor([1.0, -1.0, -1.0, -1.0])
    rotation = _quat_mul(q0_inv.expand_as(q), q)
    data = {
        "video": video,
        "translation": translation,
        "acceleration": torch.randn(B, L, 3, generator=g),
        "rotation": rotation,
        "rotation_rate": torch.randn(B, L, 3, generator=g),
        "speed": 15.0 * torch.rand(B, L, 1, generator=g),
        "temporal_offsets": torch.linspace(-0.5 * (L - 1), 0.0, L).repeat(B, 1),
        "annotated_frame_idx": torch.full((B,), L - 1, dtype=torch.int64),
    }
    boxes = torch.zeros(B, MAX_NUM_OBJECTS, 4)
    classes = torch.zeros(B, MAX_NUM_OBJECTS, dtype=torch.int64)
    active = torch.zeros(B, MAX_NUM_OBJECTS, dtype=torch.int64)
    for b in range(B):
        nb = int(torch.randint(min_boxes, max_boxes + 1, (1,), generator=g))
        wh = torch.rand(nb, 2, generator=g) * torch.tensor([W * 0.4, H * 0.4]) + 4.0
        xy = torch.rand(nb, 2, generator=g) * (torch.tensor([float(W), float(H)]) - wh)
        boxes[b, :nb] = torch.cat([xy, xy + wh], dim=1)
        classes[b, :nb] = torch.randint(0, num_classes, (nb,), generator=g)
        active[b, :nb] = 1
    data.update(boxes=boxes, classes=classes, active=active,
                ignore_boxes=torch.zeros(B, MAX_NUM_OBJECTS, 4))
    out = {k: (v.to(device) if isinstance(v, torch.Tensor) else v) for k, v in data.items()}
    # host copies of the annotations, as recursive_to() keeps them for a loader batch
    out["_host_annotations"] = {"active": active, "boxes": boxes, "classes": classes}
    return out
