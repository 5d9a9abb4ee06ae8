"""Sampling helpers (reference: caspr/models/utils.py)."""
from math import log, pi

import numpy as np
import torch


def standard_normal_logprob(z):
    log_z = -0.5 * log(2 * pi)
    return log_z - z.pow(2) / 2


def truncated_normal(tensor, mean=0, std=1, trunc_std=2):
    """models/utils.py:15-22: pick the first of 4 normal draws inside (-trunc_std, trunc_std)."""
    size = tensor.shape
    tmp = tensor.new_empty(size + (4,)).normal_()
    valid = (tmp < trunc_std) & (tmp > -trunc_std)
    ind = valid.max(-1, keepdim=True)[1]
    tensor.data.copy_(tmp.gather(-1, ind).squeeze(-1))
    tensor.data.mul_(std).add_(mean)
    return tensor


def sample_gaussian(size, truncate_std=None, device=None):
    """models/utils.py:24-29: drawn on the CPU generator, then moved (reproducible from torch.manual_seed)."""
    y = torch.randn(*size).float()
    y = y if device is None else y.to(device)
    if truncate_std is not None:
        truncated_normal(y, mean=0, std=1, trunc_std=truncate_std)
    return y


def sphere_surface_points(num_points, radius=0.5):
    """utils/transform_utils.py:80-85 (numpy global RNG, as the reference)."""
    uniform_cube = np.random.uniform(low=-1.0, high=1.0, size=(num_points, 3))
    norm_uniform = uniform_cube / np.linalg.norm(uniform_cube, axis=1).reshape((-1, 1))
    return norm_uniform * radius
