"""The six symbols `caspr/models/pointnet2.py:7` imports from Kaolin v0.1, with Kaolin's own (channels-first) signatures, on
the HIP kernels of libcaspr_hip.so -- the operator-level swap of INTEGRATION.md section 2: a maintainer of the reference
replaces that import line by

    from caspr_amd.compat.kaolin_amd import (separate_xyz_and_features, PointNet2GroupingLayer, furthest_point_sampling,
                                             fps_gather_by_index, three_nn, three_interpolate)

and keeps every PyTorch module of pointnet2.py as it is.  Index operators are bit-exact against the oracle's restatement of
the Kaolin / Pointnet2_PyTorch kernels (tests/test_hip_parity.py::test_kaolin_compat_*); the three operators that carry a
gradient in Kaolin (gather, grouping, three_interpolate) are torch.autograd.Functions over the backward entries of
include/caspr_hip_train.h.  GPU tensors only: there is no CPU path.

(The model-level swap -- `caspr_amd.models.CaSPR` -- does not go through this module: it keeps activations point-major and
fuses the grouper into the set-abstraction kernel.  This module is the thin route for callers that want the reference's
own PointNet++ code.)"""
import torch
import torch.nn as nn

from .. import ops, train_ops


def _rows(t_cf):
    """(B,C,P) channels-first -> point-major (B,P,roundup4(C)) rows (what the C ABI takes), zero padded."""
    B, C, P = t_cf.shape
    ld = (C + 3) // 4 * 4
    r = torch.zeros(B, P, ld, device=t_cf.device, dtype=torch.float32)
    r[:, :, :C] = t_cf.transpose(1, 2)
    return r


def separate_xyz_and_features(points):
    """(B,n,3+C) -> xyz (B,n,3), features (B,C,n) or None (pointnet2.py:228)."""
    xyz = points[..., 0:3].contiguous()
    features = points[..., 3:].transpose(1, 2).contiguous() if points.shape[-1] > 3 else None
    return xyz, features


def furthest_point_sampling(xyz, num_points_out):
    """(B,n,3) -> (B,num_points_out) int32 (pointnet2.py:384).  No gradient (indices)."""
    return ops.furthest_point_sampling(xyz.detach().contiguous(), int(num_points_out))


class _GatherByIndex(torch.autograd.Function):
    @staticmethod
    def forward(ctx, features, idx):
        ctx.save_for_backward(idx)
        ctx.n = features.shape[2]
        C = features.shape[1]
        return ops.gather_points(_rows(features), idx.contiguous())[:, :, :C].transpose(1, 2).contiguous()

    @staticmethod
    def backward(ctx, g):            # (B,C,M) -> (B,C,n): scatter-add (an index may repeat)
        (idx,) = ctx.saved_tensors
        d = torch.zeros(g.shape[0], g.shape[1], ctx.n, device=g.device, dtype=g.dtype)
        d.scatter_add_(2, idx.long().unsqueeze(1).expand(-1, g.shape[1], -1), g)
        return d, None


def fps_gather_by_index(features, idx):
    """features (B,C,n), idx (B,M) -> (B,C,M) (pointnet2.py:385)."""
    return _GatherByIndex.apply(features, idx)


class _Group(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xyz, new_xyz, features, idx):
        C = 0 if features is None else features.shape[1]
        feat = None if features is None else _rows(features)
        out = ops.group_points(xyz.contiguous(), new_xyz.contiguous(), feat, idx)         # (B,M,3+ld,ns): xyz rows first
        if feat is not None and feat.shape[2] != C:
            out = out[:, :, :3 + C].contiguous()
        ctx.save_for_backward(idx)
        ctx.C, ctx.n = C, xyz.shape[1]
        return out

    @staticmethod
    def backward(ctx, g):            # (B,M,3+C,ns) -> d features (B,C,n); xyz carries no gradient in the reference's use
        (idx,) = ctx.saved_tensors
        C = ctx.C
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            # Kaolin's grouper is differentiable in the coordinates (scatter of the first three rows, minus their sum for new_xyz);
            # the reference never asks for that (its clouds are data) and this build has no kernel for it: refuse rather than
            # hand back a silent None
            raise NotImplementedError("PointNet2GroupingLayer: the gradient with respect to xyz / new_xyz is not built "
                                      "(detach the coordinates; the reference's inputs do not require grad)")
        if C == 0 or not ctx.needs_input_grad[2]:
            return None, None, None, None
        B, M, _, ns = g.shape
        ld = (3 + C + 3) // 4 * 4
        rows = torch.zeros(B, M * ns, ld, device=g.device, dtype=torch.float32)
        rows[:, :, :3 + C] = g.permute(0, 1, 3, 2).reshape(B, M * ns, 3 + C)
        d = torch.zeros(B, ctx.n, (C + 3) // 4 * 4, device=g.device, dtype=torch.float32)
        train_ops.group_rows_bwd(rows, idx, C, d)
        return None, None, d[:, :, :C].transpose(1, 2).contiguous(), None


class PointNet2GroupingLayer(nn.Module):
    """Kaolin's grouper (pointnet2.py:340-342, 391-398): ball query around `new_xyz`, gather, centre ->
    (B, M, 3 + C, num_samples), xyz rows first (`use_xyz_feature=True`, the only form the reference builds)."""

    def __init__(self, radius, num_samples, use_xyz_feature=True, use_random_ball_query=False):
        super().__init__()
        if use_random_ball_query:
            raise NotImplementedError("use_random_ball_query=True is not built (the reference passes False, tpointnet2.py:45-52)")
        if not use_xyz_feature:
            raise NotImplementedError("use_xyz_feature=False is not built (the reference passes True, tpointnet2.py:45-52)")
        self.radius, self.num_samples = radius, num_samples

    def forward(self, xyz, new_xyz, features=None):
        if new_xyz is None:
            raise NotImplementedError("group-all (new_xyz=None) is not built: every set abstraction of the reference samples centres")
        idx = ops.ball_query(self.radius, self.num_samples, xyz.detach().contiguous(), new_xyz.detach().contiguous())
        return _Group.apply(xyz, new_xyz, features, idx)


def three_nn(unknown, known):
    """(B,n,3), (B,m,3) -> dist (B,n,3) [square roots], idx (B,n,3) int32 (pointnet2.py:514).  No gradient."""
    return ops.three_nn(unknown.detach().contiguous(), known.detach().contiguous())


class _ThreeInterpolate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, features, idx, weight):
        C = features.shape[1]
        ctx.save_for_backward(idx, weight)
        ctx.m, ctx.C = features.shape[2], C
        rows = _rows(features)                               # the kernel interpolates whole float4 columns: the zero padding rides along
        out = ops.three_interpolate(rows, idx.contiguous(), weight.contiguous(), C=rows.shape[2])
        return out[:, :, :C].transpose(1, 2).contiguous()

    @staticmethod
    def backward(ctx, g):            # (B,C,n) -> (B,C,m)
        idx, weight = ctx.saved_tensors
        d = torch.zeros(g.shape[0], ctx.m, (ctx.C + 3) // 4 * 4, device=g.device, dtype=torch.float32)
        train_ops.three_interp_bwd(_rows(g), idx, weight.contiguous(), d.shape[2], d)
        return d[:, :, :ctx.C].transpose(1, 2).contiguous(), None, None


def three_interpolate(features, idx, weight):
    """features (B,C,m), idx (B,n,3), weight (B,n,3) -> (B,C,n) (pointnet2.py:519)."""
    return _ThreeInterpolate.apply(features, idx, weight)
