"""Deterministic synthetic inputs and weights for tests / bench.

No dataset or checkpoint can be fetched here, so both are generated: "car-like" rigid sequences
(points on the surface of a rotating box at depth ~2.2, as data/demo's depth range) with the
reference's timestamp conventions (caspr_dataset.py:200-204), and a seeded parameter stream that is
a function of (seed, state_dict key) only."""
import zlib

import numpy as np
import torch


def seeded_state_dict(reference_sd, seed=0):
    """Return a new state dict with the keys/shapes of `reference_sd` and deterministic values."""
    out = {}
    for k, v in reference_sd.items():
        rng = np.random.default_rng([seed, zlib.crc32(k.encode())])
        shape = tuple(v.shape)
        name = k.split('.')[-1]
        if name in ("_num_evals", "step"):
            a = np.zeros(shape)
        elif name == "sqrt_end_time":
            a = np.full(shape, np.sqrt(0.5))
        elif name == "running_mean":
            a = rng.normal(0, 0.1, shape)
        elif name == "running_var":
            a = rng.uniform(0.5, 1.5, shape)
        elif "point_cnf.chain" in k and k.count('.') == 3 and name in ("weight", "bias"):   # MovingBatchNorm affine
            a = rng.normal(0, 0.1, shape)
        elif "dynamics_net" in k:
            a = rng.normal(0, 0.1, shape) if name == "weight" else rng.normal(0, 0.01, shape)
        elif name == "weight" and len(shape) >= 2:
            fan_in = int(np.prod(shape[1:]))
            a = rng.normal(0, 1.0 / np.sqrt(fan_in), shape)
        elif name == "weight":            # GroupNorm gamma
            a = 1.0 + 0.1 * rng.normal(0, 1, shape)
        else:                             # biases / GroupNorm beta
            a = rng.normal(0, 0.05, shape)
        out[k] = torch.from_numpy(np.asarray(a, dtype=np.float32).reshape(shape))
    # the dynamics net is registered twice (latent_ode.ode_func.* and latent_ode.solver.ode_func.*): keep the aliases equal
    for k in list(out):
        if k.startswith("latent_ode.solver.ode_func."):
            out[k] = out["latent_ode.ode_func." + k[len("latent_ode.solver.ode_func."):]].clone()
    return out


def car_sequences(B, T, N, seed=1234, max_timestamp=5.0):
    """-> x (B,T,N,4) camera-frame points + time in [0,max_timestamp];
          sample_points (B,T,N,4) the same points in the unit NOCS cube + time in [0,1]."""
    x = np.zeros((B, T, N, 4), np.float32)
    s = np.zeros((B, T, N, 4), np.float32)
    half = np.array([0.45, 0.2, 0.15])
    for b in range(B):
        rng = np.random.default_rng(seed + b)
        u = rng.uniform(0.3, 1.0)
        for t in range(T):
            face = rng.integers(0, 6, N)
            p = rng.uniform(-1, 1, (N, 3))
            ax = face % 3
            p[np.arange(N), ax] = np.where(face < 3, 1.0, -1.0)
            p = p * half
            nocs = p / (2 * half.max()) + 0.5
            ang = 2 * np.pi * u * t / max(T, 1)
            R = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]])
            cam = p @ R.T + np.array([0.05 * t, 0.0, 2.2])
            tt = t / max(T - 1, 1)
            x[b, t, :, :3] = cam
            x[b, t, :, 3] = max_timestamp * tt
            s[b, t, :, :3] = nocs
            s[b, t, :, 3] = tt
    return torch.from_numpy(x), torch.from_numpy(s)


def random_clouds(B, T, N, seed=1234, max_timestamp=5.0):
    """i.i.d. U(0,1)^3 clouds shifted to depth (config 5 of BASELINE.json)."""
    rng = np.random.default_rng(seed)
    x = np.zeros((B, T, N, 4), np.float32)
    x[..., :3] = rng.uniform(0, 1, (B, T, N, 3)) + np.array([0, 0, 1.5])
    x[..., 3] = (max_timestamp * np.arange(T) / max(T - 1, 1))[None, :, None]
    return torch.from_numpy(x)


def dense_sequences(B, T, N, seed=4321, side=0.12, max_timestamp=5.0):
    """Well-conditioned parity input: uniform points in a small cube (side 0.12 at depth 2.2) so that every
    r=0.02 ball of the first set-abstraction level holds >= 16 distinct points.  (On sparse clouds most
    neighbourhoods are padded with duplicates and GroupNorm over near-constant samples amplifies f32
    rounding by up to 1/sqrt(eps) = 316x -- in the reference as much as here; see DESIGN.md.)"""
    rng = np.random.default_rng(seed)
    x = np.zeros((B, T, N, 4), np.float32)
    x[..., :3] = rng.uniform(-side / 2, side / 2, (B, T, N, 3)) + np.array([0.02, 0.01, 2.2])
    x[..., 3] = (max_timestamp * np.arange(T) / max(T - 1, 1))[None, :, None]
    s = np.zeros((B, T, N, 4), np.float32)
    s[..., :3] = (x[..., :3] - np.array([0.02, 0.01, 2.2])) / side + 0.5
    s[..., 3] = (np.arange(T) / max(T - 1, 1))[None, :, None]
    return torch.from_numpy(x), torch.from_numpy(s)
