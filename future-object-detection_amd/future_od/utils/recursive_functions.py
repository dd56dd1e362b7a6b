import torch


def _walk(data, fn):
    if isinstance(data, dict):
        return {k: _walk(v, fn) for k, v in data.items()}
    if isinstance(data, (list, tuple)):
        return [_walk(v, fn) for v in data]
    if isinstance(data, torch.Tensor):
        return fn(data)
    return data


def recursive_detach_cpu(data):
    return _walk(data, lambda t: t.detach())


HOST_ANNOTATIONS = "_host_annotations"
_ANNOTATION_KEYS = ("active", "boxes", "classes")


def recursive_to(data, device):
    """Move a batch to `device`.  The loader's HOST copies of the dense annotation tensors are kept under
    `_host_annotations`: the set loss needs the per-sample target counts on the host, and reading them back from
    the device would stall the host on everything queued so far (the previous step's backward) at the very start
    of each step."""
    out = _walk(data, lambda t: t.to(device, non_blocking=True))
    if (isinstance(data, dict) and HOST_ANNOTATIONS not in data and all(k in data for k in _ANNOTATION_KEYS)
            and all(isinstance(data[k], torch.Tensor) and data[k].device.type == "cpu" for k in _ANNOTATION_KEYS)):
        out[HOST_ANNOTATIONS] = {k: data[k] for k in _ANNOTATION_KEYS}
    return out


def recursive_tensor_sizes(data):
    return _walk(data, lambda t: t.size())
