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


def recursive_to(data, device):
    return _walk(data, lambda t: t.to(device))


def recursive_tensor_sizes(data):
    return _walk(data, lambda t: t.size())
