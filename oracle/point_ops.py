"""oracle/point_ops.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

ctypes front-end of oracle/point_ops.c (the CPU restatement of the Kaolin / tk3dv operators the
reference calls at caspr/models/pointnet2.py:7,384-391,514-519 and caspr/utils/evaluations.py:40).
PARITY UNPINNED for these third-party operators -- see the header of point_ops.c.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes
import os
import subprocess

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle_point_ops.so")
_lib = None


def build(force=False):
    """Compile point_ops.c with gcc (recipe: oracle/Makefile)."""
    src = os.path.join(_HERE, "point_ops.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
    return _lib


def _f(t):
    assert t.dtype == torch.float32 and t.is_contiguous() and t.device.type == "cpu"
    return ctypes.c_void_p(t.data_ptr())


def _i(t):
    assert t.dtype == torch.int32 and t.is_contiguous() and t.device.type == "cpu"
    return ctypes.c_void_p(t.data_ptr())


def furthest_point_sampling(xyz, M, guard=True):
    xyz = xyz.detach().float().contiguous()
    B, n, _ = xyz.shape
    idx = torch.zeros(B, M, dtype=torch.int32)
    lib().oracle_fps(_f(xyz), B, n, M, int(bool(guard)), _i(idx))
    return idx


def _torch_path(*ts):
    """Differentiable / f64 route: plain torch indexing instead of the C library (same values; the C calls detach).
    Taken when any operand is float64 (conditioning studies) or takes part in autograd (gradient fixtures)."""
    return any(t is not None and (t.dtype == torch.float64 or (torch.is_grad_enabled() and t.requires_grad)) for t in ts)


def _bidx(idx):
    return torch.arange(idx.shape[0]).view(-1, *([1] * (idx.dim() - 1))).expand_as(idx)


def fps_gather_by_index(feat, idx):
    if _torch_path(feat):
        return torch.gather(feat, 2, idx.long().unsqueeze(1).expand(-1, feat.shape[1], -1))
    feat = feat.detach().float().contiguous()
    idx = idx.contiguous()
    B, C, n = feat.shape
    M = idx.shape[1]
    out = torch.empty(B, C, M, dtype=torch.float32)
    lib().oracle_gather(_f(feat), _i(idx), B, C, n, M, _f(out))
    return out


def ball_query(radius, ns, xyz, new_xyz):
    xyz = xyz.detach().float().contiguous()
    new_xyz = new_xyz.detach().float().contiguous()
    B, n, _ = xyz.shape
    M = new_xyz.shape[1]
    idx = torch.zeros(B, M, ns, dtype=torch.int32)
    lib().oracle_ball_query(_f(xyz), _f(new_xyz), B, n, M, ctypes.c_float(float(np.float32(radius))), ns, _i(idx))
    return idx


def group(xyz, new_xyz, feat, idx):
    """-> (B, M, 3+C, ns): centred xyz rows first, then feature rows."""
    if _torch_path(xyz, feat):
        li = idx.long()
        g = xyz[_bidx(li), li] - new_xyz.unsqueeze(2)                      # (B,M,ns,3)
        g = g.permute(0, 1, 3, 2)
        if feat is not None:
            f = feat.transpose(1, 2)[_bidx(li), li].permute(0, 1, 3, 2)    # (B,M,C,ns)
            g = torch.cat([g, f], dim=2)
        return g.contiguous()
    xyz = xyz.detach().float().contiguous()
    new_xyz = new_xyz.detach().float().contiguous()
    B, n, _ = xyz.shape
    M, ns = idx.shape[1], idx.shape[2]
    C = 0 if feat is None else feat.shape[1]
    out = torch.empty(B, M, 3 + C, ns, dtype=torch.float32)
    fp = ctypes.c_void_p(0) if feat is None else _f(feat.detach().float().contiguous())
    lib().oracle_group(_f(xyz), _f(new_xyz), fp, _i(idx.contiguous()), B, n, M, C, ns, _f(out))
    return out


def three_nn(unknown, known):
    unknown = unknown.detach().float().contiguous()
    known = known.detach().float().contiguous()
    B, n, _ = unknown.shape
    m = known.shape[1]
    dist = torch.empty(B, n, 3, dtype=torch.float32)
    idx = torch.empty(B, n, 3, dtype=torch.int32)
    lib().oracle_three_nn(_f(unknown), _f(known), B, n, m, _f(dist), _i(idx))
    return dist, idx


def three_nn_f64(unknown, known, idx):
    """Distances of the (f32-selected) neighbours recomputed in f64 (conditioning studies only)."""
    li = idx.long()
    d = unknown.unsqueeze(2) - known[_bidx(li), li]
    return d.pow(2).sum(-1).sqrt()


def three_interpolate(feat, idx, weight):
    if _torch_path(feat, weight):
        li = idx.long()
        f = feat.transpose(1, 2)[_bidx(li), li]                            # (B,n,3,C)
        return (f * weight.unsqueeze(-1)).sum(dim=2).transpose(1, 2).contiguous()
    feat = feat.detach().float().contiguous()
    weight = weight.detach().float().contiguous()
    B, C, m = feat.shape
    n = idx.shape[1]
    out = torch.empty(B, C, n, dtype=torch.float32)
    lib().oracle_three_interp(_f(feat), _i(idx.contiguous()), _f(weight), B, C, m, n, _f(out))
    return out


def chamfer(p, q):
    p = p.detach().float().contiguous()
    q = q.detach().float().contiguous()
    B, n, _ = p.shape
    m = q.shape[1]
    d1 = torch.empty(B, n, dtype=torch.float32)
    d2 = torch.empty(B, m, dtype=torch.float32)
    lib().oracle_chamfer(_f(p), _f(q), B, n, m, _f(d1), _f(d2))
    return d1, d2


def separate_xyz_and_features(points):
    """Kaolin helper used at pointnet2.py:228: (B,n,3+C) -> xyz (B,n,3), feat (B,C,n) | None."""
    xyz = points[..., 0:3].contiguous()
    feat = points[..., 3:].transpose(1, 2).contiguous() if points.shape[-1] > 3 else None
    return xyz, feat
