"""Training tier: the HIP gradient kernels (include/caspr_hip_train.h) against torch.autograd evaluated
in float64 on the CPU over the same graph (the reference obtains these gradients from autograd,
train_utils.py:173).  Tolerances are relative to the largest reference entry of each tensor: the
kernels accumulate in f32 (MFMA) / f64 (statistics), rtol 2e-5 unless noted.
"""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
REPORT = {}


def rnd(seed, *shape, scale=1.0):
    return torch.from_numpy((np.random.default_rng(seed).normal(0, 1, shape) * scale).astype(np.float32))


def rel(name, got, want, rtol):
    got = got.detach().cpu().double().numpy()
    want = want.detach().cpu().double().numpy()
    assert got.shape == want.shape, "%s: shape %s vs %s" % (name, got.shape, want.shape)
    ref = float(np.abs(want).max()) or 1.0
    err = float(np.abs(got - want).max()) / ref
    REPORT[name] = {"max_rel_err": err, "rtol": rtol, "ref_absmax": ref}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "train_parity_report.json"), "w") as f:
        json.dump(REPORT, f, indent=1, sort_keys=True)
    assert np.isfinite(got).all(), "%s: non-finite" % name
    assert err <= rtol, "%s: max err / |ref|max = %.3e > %.1e" % (name, err, rtol)


def pad4(t):
    c = t.shape[-1]
    return F.pad(t, (0, (-c) % 4)) if c % 4 else t


@pytest.mark.parametrize("B,P,Cin,C1,C2", [(3, 1500, 7, 64, 130), (2, 2048, 64, 128, 64), (1, 333, 4, 256, 3)])
def test_conv_gn_relu_block_backward(B, P, Cin, C1, C2):
    """x -> conv(W1,b1) -> GN(16)+ReLU -> conv(W2,b2) -> sum(. * R): every parameter gradient and dx."""
    from caspr_amd import ops, train_ops as T
    dev = "cuda:0"
    x, R = rnd(1, B, P, Cin), rnd(2, B, P, C2)
    W1, b1 = rnd(3, C1, Cin, scale=0.3), rnd(4, C1, scale=0.1)
    g1, be1 = 1 + rnd(5, C1, scale=0.2), rnd(6, C1, scale=0.2)
    W2, b2 = rnd(7, C2, C1, scale=0.1), rnd(8, C2, scale=0.1)
    # reference: float64 autograd
    p64 = [t.double().requires_grad_(True) for t in (x, W1, b1, g1, be1, W2, b2)]
    x6, W16, b16, g16, be16, W26, b26 = p64
    y1 = x6 @ W16.t() + b16
    a1 = F.relu(F.group_norm(y1.transpose(1, 2), 16, g16, be16, 1e-5)).transpose(1, 2)
    y2 = a1 @ W26.t() + b26
    (y2 * R.double()).sum().backward()
    # HIP
    xin = pad4(x).to(dev)[:, :, :Cin]
    W1d, b1d, g1d, be1d, W2d, b2d = (t.to(dev) for t in (W1, b1, g1, be1, W2, b2))
    pw1, pw2 = ops.PackedWeight(W1d), ops.PackedWeight(W2d)
    y1d = ops.conv1x1(pw1, b1d, xin)
    sc, sh, mean, rstd = T.gn_stats_train(y1d, C1, g1d, be1d)
    y2d = ops.conv1x1(pw2, b2d, y1d, in_scale=sc, in_shift=sh, in_relu=True)
    rel("fwd_y2[%d,%d]" % (C1, C2), y2d[:, :, :C2], y2, 2e-5)
    dy2 = pad4(R).to(dev).contiguous()
    dW2, db2 = torch.empty_like(W2d), torch.empty_like(b2d)
    T.conv1x1_wgrad(dy2, y1d, C1, C2, dW2, db2, in_scale=sc, in_shift=sh, in_relu=True)
    da1 = ops.conv1x1(ops.PackedWeight(W2d.t().contiguous()), None, dy2)
    dg1, dbe1 = torch.empty_like(g1d), torch.empty_like(be1d)
    T.gn_bwd(y1d, da1, C1, mean, rstd, g1d, be1d, dg1, dbe1, relu=True)
    dW1, db1 = torch.empty_like(W1d), torch.empty_like(b1d)
    T.conv1x1_wgrad(da1, xin, Cin, C1, dW1, db1)
    dx = ops.conv1x1(ops.PackedWeight(W1d.t().contiguous()), None, da1)
    tag = "[%d,%d,%d]" % (Cin, C1, C2)
    rel("dW2" + tag, dW2, W26.grad, 2e-5)
    rel("db2" + tag, db2, b26.grad, 2e-5)
    rel("dgamma1" + tag, dg1, g16.grad, 5e-5)
    rel("dbeta1" + tag, dbe1, be16.grad, 5e-5)
    rel("dW1" + tag, dW1, W16.grad, 5e-5)
    rel("db1" + tag, db1, b16.grad, 5e-4)   # mathematically ~0 after GroupNorm: compare against |dW1| scale instead
    rel("dx" + tag, dx[:, :, :Cin], x6.grad, 5e-5)
    # accumulate flag adds onto existing contents; repeated calls are bit-identical (fixed-order reductions)
    dW2b = dW2.clone()
    T.conv1x1_wgrad(dy2, y1d, C1, C2, dW2b, None, in_scale=sc, in_shift=sh, in_relu=True, accumulate=True)
    rel("dW2_accumulate" + tag, dW2b, 2 * W26.grad, 2e-5)
    dW2c = torch.empty_like(W2d)
    T.conv1x1_wgrad(dy2, y1d, C1, C2, dW2c, None, in_scale=sc, in_shift=sh, in_relu=True)
    assert torch.equal(dW2c, dW2)
