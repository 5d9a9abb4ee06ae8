"""Base-distribution helpers of the decoder.  Behavioural contract = caspr/models/utils.py:10-29 (the reference
draws on the CPU generator and then moves the tensor, so `torch.manual_seed` reproduces its samples exactly) and
caspr/utils/transform_utils.py:80-85 (`sphere_surface_points`, numpy global RNG)."""
import math

import numpy as np
import torch

_LOG_SQRT_2PI = 0.5 * math.log(2.0 * math.pi)


def standard_normal_logprob(z):
    """Element-wise log N(z; 0, 1)."""
    return -_LOG_SQRT_2PI - 0.5 * z * z


def truncated_normal(tensor, mean=0, std=1, trunc_std=2):
    """In-place: each element becomes the first of four fresh N(0,1) candidates that lies inside
    (-trunc_std, trunc_std) (candidate 0 if none does), then is scaled and shifted."""
    candidates = tensor.new_empty(tuple(tensor.shape) + (4,)).normal_()
    inside = candidates.abs() < trunc_std
    first = inside.to(torch.uint8).argmax(dim=-1, keepdim=True)     # argmax returns the first maximal index
    picked = torch.gather(candidates, -1, first).squeeze(-1)
    tensor.data.copy_(picked * std + mean)
    return tensor


def sample_gaussian(size, truncate_std=None, device=None):
    """N(0,1) samples of shape `size`, drawn with the CPU generator and moved to `device` afterwards."""
    y = torch.randn(*size, dtype=torch.float32)
    if device is not None:
        y = y.to(device)
    if truncate_std is not None:
        truncated_normal(y, mean=0, std=1, trunc_std=truncate_std)
    return y


def sphere_surface_points(num_points, radius=0.5):
    """Uniform directions (normalised uniform-cube draws, as the reference) on a sphere of the given radius."""
    cube = np.random.uniform(low=-1.0, high=1.0, size=(num_points, 3))
    return radius * cube / np.linalg.norm(cube, axis=1, keepdims=True)
