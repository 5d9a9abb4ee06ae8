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


def rel(name, got, want, rtol, ref=None):
    """max |got - want| <= rtol * ref, ref = largest reference entry (or a given scale, for gradients that are
    mathematically zero such as a conv bias in front of a one-channel-per-group GroupNorm)."""
    got = got.detach().cpu().double().numpy()
    want = want.detach().cpu().double().numpy()
    assert got.shape == want.shape, "%s: shape %s vs %s" % (name, got.shape, want.shape)
    ref = float(ref) if ref is not None else (float(np.abs(want).max()) or 1.0)
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


@pytest.fixture(params=["bf16x6", "f32"])
def matmul_mode(request):
    """Both kernel families of the matrix products (ops.set_matmul_mode), same thresholds."""
    from caspr_amd import ops
    prev = ops.set_matmul_mode(request.param)
    yield request.param
    ops.set_matmul_mode(conv=prev[0], cnf=prev[1])


@pytest.mark.parametrize("B,P,Cin,C1,C2", [(3, 1500, 7, 64, 130), (2, 2048, 64, 128, 64), (1, 333, 4, 256, 3)])
def test_conv_gn_relu_block_backward(matmul_mode, B, P, Cin, C1, C2):
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


@pytest.mark.parametrize("B,P,Cin,Cout", [(2, 1024, 608, 512), (1, 4100, 1600, 1600), (3, 700, 132, 260), (2, 96, 512, 64)])
def test_conv1x1_wgrad_direct(matmul_mode, B, P, Cin, Cout):
    """dW = dY^T . relu(x * scale + shift), db = column sums of dY, against float64 -- both kernel families at the same
    tolerance (the bf16x6 one: both operands split exactly on the way to LDS, fragments through ds_read_b64_tr_b16), ragged
    channel counts and row counts, the fused input transform, bit-reproducibility."""
    from caspr_amd import train_ops as T
    dev = "cuda:0"
    x, dy = rnd(1, B, P, Cin), rnd(2, B, P, Cout)
    sc, sh = rnd(3, B, Cin).abs() + 0.5, rnd(4, B, Cin)
    xin = torch.relu(x.double() * sc.double().unsqueeze(1) + sh.double().unsqueeze(1))
    want = torch.einsum("bpo,bpi->oi", dy.double(), xin)
    xd, dyd = x.to(dev), dy.to(dev)
    dW, db = torch.empty(Cout, Cin, device=dev), torch.empty(Cout, device=dev)
    T.conv1x1_wgrad(dyd, xd, Cin, Cout, dW, db, in_scale=sc.to(dev), in_shift=sh.to(dev), in_relu=True)
    rel("wgrad_fused[%d,%d]" % (Cin, Cout), dW, want, 3e-6)
    rel("wgrad_bias[%d,%d]" % (Cin, Cout), db, dy.double().sum(dim=(0, 1)), 3e-6)
    dW2 = torch.empty_like(dW)
    T.conv1x1_wgrad(dyd, xd, Cin, Cout, dW2, None, in_scale=sc.to(dev), in_shift=sh.to(dev), in_relu=True)
    assert torch.equal(dW, dW2)
    T.conv1x1_wgrad(dyd, xd, Cin, Cout, dW2, None)
    rel("wgrad_plain[%d,%d]" % (Cin, Cout), dW2, torch.einsum("bpo,bpi->oi", dy.double(), x.double()), 3e-6)


@pytest.mark.parametrize("B,P,Cin,Cout", [(2, 40000, 9, 32), (1, 70001, 16, 16), (3, 30000, 32, 64), (2, 33000, 99, 32), (1, 66000, 131, 64),
                                          (2, 35000, 96, 128), (1, 65536, 64, 96), (1, 65600, 3, 5), (2, 33000, 160, 17), (2, 40000, 64, 32), (1, 70000, 16, 128),
                                          (1, 70000, 32, 100), (2, 36000, 64, 64)])
def test_conv1x1_wgrad_narrow(matmul_mode, B, P, Cin, Cout):
    """The weight gradient of a narrow conv over many rows (the training encoder's set-abstraction MLPs: conv1x1_wgrad_narrow_kernel, no
    LDS stage, operands straight from global memory): against float64 with the bias gradient, garbage in the pad columns of both
    operands, accumulate, bit-reproducibility -- every tile-count instantiation the dispatch can pick."""
    from caspr_amd import train_ops as T
    dev = "cuda:0"
    ldx, ldy = (Cin + 3) // 4 * 4 + 4, (Cout + 3) // 4 * 4
    xw, dyw = rnd(1, B, P, ldx), rnd(2, B, P, ldy)
    x, dy = xw[:, :, :Cin], dyw[:, :, :Cout]
    want = torch.einsum("bpo,bpi->oi", dy.double(), x.double())
    xd, dyd = xw.to(dev), dyw.to(dev)
    dW, db = torch.empty(Cout, Cin, device=dev), torch.empty(Cout, device=dev)
    T.conv1x1_wgrad(dyd, xd, Cin, Cout, dW, db)
    rel("wgrad_narrow[%d,%d]" % (Cin, Cout), dW, want, 3e-6)
    rel("wgrad_narrow_bias[%d,%d]" % (Cin, Cout), db, dy.double().sum(dim=(0, 1)), 3e-6)
    dW2, db2 = dW.clone(), db.clone()
    T.conv1x1_wgrad(dyd, xd, Cin, Cout, dW2, db2, accumulate=True)
    rel("wgrad_narrow_accumulate[%d,%d]" % (Cin, Cout), dW2, 2 * want, 3e-6)
    rel("wgrad_narrow_bias_accumulate[%d,%d]" % (Cin, Cout), db2, 2 * dy.double().sum(dim=(0, 1)), 3e-6)
    dW3 = torch.empty_like(dW)
    T.conv1x1_wgrad(dyd, xd, Cin, Cout, dW3, None)
    assert torch.equal(dW, dW3)


@pytest.mark.parametrize("B,n,M,ns,C,dims", [(2, 256, 64, 16, 6, (16, 16, 32)), (2, 128, 32, 32, 96, (64, 96, 128)), (1, 64, 16, 32, 512, (256, 256, 512))])
def test_set_abstraction_scale_backward(B, n, M, ns, C, dims):
    """group -> 3 x (conv -> per-neighbourhood GroupNorm(16) [-> ReLU]) -> max over samples  (pointnet2.py:391-409,649-703):
    forward output, all parameter gradients and the gradient w.r.t. the input features."""
    from caspr_amd import ops, train_ops as T
    dev = "cuda:0"
    g = np.random.default_rng(5)
    xyz = torch.from_numpy(g.uniform(0, 1, (B, n, 3)).astype(np.float32))
    feat = rnd(11, B, n, C)
    idx = torch.from_numpy(g.integers(0, n, (B, M, ns)).astype(np.int32))
    ctr = xyz[:, :M].contiguous()
    R = rnd(12, B, M, dims[2])
    cin = [3 + C, dims[0], dims[1]]
    Ws = [rnd(20 + l, dims[l], cin[l], scale=0.4 / np.sqrt(cin[l]) * 3) for l in range(3)]
    bs = [rnd(30 + l, dims[l], scale=0.1) for l in range(3)]
    gs = [1 + rnd(40 + l, dims[l], scale=0.2) for l in range(3)]
    bes = [rnd(50 + l, dims[l], scale=0.2) for l in range(3)]
    # reference (f64 autograd)
    f6 = feat.double().requires_grad_(True)
    P6 = [[t.double().requires_grad_(True) for t in grp] for grp in (Ws, bs, gs, bes)]
    li = idx.long()
    bi = torch.arange(B).view(B, 1, 1)
    grouped = torch.cat([xyz.double()[bi, li] - ctr.double().unsqueeze(2), f6[bi, li]], dim=3)      # (B,M,ns,3+C)
    h = grouped.reshape(B * M, ns, 3 + C)
    for l in range(3):
        h = h @ P6[0][l].t() + P6[1][l]
        h = F.group_norm(h.transpose(1, 2), 16, P6[2][l], P6[3][l], 1e-5).transpose(1, 2)
        if l < 2:
            h = F.relu(h)
    out6 = h.max(dim=1)[0].view(B, M, dims[2])
    (out6 * R.double()).sum().backward()
    # HIP
    ldf = (C + 3) // 4 * 4
    featd = F.pad(feat, (0, ldf - C)).to(dev)
    xyzd, ctrd, idxd = xyz.to(dev), ctr.to(dev), idx.to(dev)
    Wd, bd, gd, bed = ([t.to(dev) for t in grp] for grp in (Ws, bs, gs, bes))
    G = T.group_rows(xyzd, ctrd, featd, C, idxd)
    tape, cur = [], G
    out = torch.empty(B, M, dims[2], device=dev)
    for l in range(3):
        y = ops.conv1x1(ops.PackedWeight(Wd[l]), bd[l], cur)
        A, mean, rstd, arg = T.gn_rows(y, ns, dims[l], gd[l], bed[l], relu=l < 2, maxout=out if l == 2 else None)
        tape.append((cur, y, mean, rstd, arg))
        cur = A
    tag = "[%d,%d,%s]" % (C, ns, "-".join(map(str, dims)))
    rel("sa_fwd" + tag, out, out6, 2e-5)
    dout = R.to(dev).contiguous()
    d = None
    for l in (2, 1, 0):
        xin, y, mean, rstd, arg = tape[l]
        dg, dbe = torch.empty_like(gd[l]), torch.empty_like(bed[l])
        if l == 2:
            dy = T.gn_rows_bwd(y, ns, dims[l], gd[l], bed[l], False, mean, rstd, dg, dbe, dmax=dout, arg=arg)
        else:
            dy = T.gn_rows_bwd(y, ns, dims[l], gd[l], bed[l], True, mean, rstd, dg, dbe, da=d)
        dW, db = torch.empty_like(Wd[l]), torch.empty_like(bd[l])
        T.conv1x1_wgrad(dy, xin, cin[l], dims[l], dW, db)
        d = ops.conv1x1(ops.PackedWeight(Wd[l].t().contiguous()), None, dy)
        rel("sa_dW%d" % l + tag, dW, P6[0][l].grad, 1e-4)
        rel("sa_db%d" % l + tag, db, P6[1][l].grad, 1e-4, ref=float(P6[0][l].grad.abs().max()))
        rel("sa_dgamma%d" % l + tag, dg, P6[2][l].grad, 1e-4)
        rel("sa_dbeta%d" % l + tag, dbe, P6[3][l].grad, 1e-4)
    dfeat = torch.zeros(B, n, ldf, device=dev)
    T.group_rows_bwd(d, idxd, C, dfeat)
    rel("sa_dfeat" + tag, dfeat[:, :, :C], f6.grad, 1e-4)


def test_three_interp_backward():
    from caspr_amd import ops, train_ops as T
    dev = "cuda:0"
    B, m, n, C = 3, 64, 256, 96
    g = np.random.default_rng(6)
    known, unknown = rnd(1, B, m, 3), rnd(2, B, n, 3)
    feat, R = rnd(3, B, m, C), rnd(4, B, n, C)
    _, idx, w = ops.three_nn(unknown.to(dev), known.to(dev), with_weights=True)
    f6 = feat.double().requires_grad_(True)
    bi = torch.arange(B).view(B, 1, 1)
    interp = (f6[bi, idx.cpu().long()] * w.cpu().double().unsqueeze(3)).sum(2)
    (interp * R.double()).sum().backward()
    dfeat = torch.zeros(B, m, C, device=dev)
    T.three_interp_bwd(R.to(dev), idx, w, C, dfeat)
    rel("three_interp_bwd", dfeat, f6.grad, 1e-5)


def test_head_max_backward():
    """z = max_p GN(conv(x)) (no ReLU, tpointnet2.py:100,111) and t = conv2(relu(GN(conv(x)))): both consumers of one GroupNorm."""
    from caspr_amd import ops, train_ops as T
    dev = "cuda:0"
    B, P, Cin, C1, C2 = 3, 2500, 32, 64, 4
    x, Rz, Rt = rnd(1, B, P, Cin), rnd(2, B, C1), rnd(3, B, P, C2)
    W1, b1, g1, be1 = rnd(4, C1, Cin, scale=0.3), rnd(5, C1, scale=0.1), 1 + rnd(6, C1, scale=0.3), rnd(7, C1, scale=0.2)
    g1[::5] *= -1   # negative gamma: the max of the normalised feature is the min of the raw one
    W2, b2 = rnd(8, C2, C1, scale=0.2), rnd(9, C2, scale=0.1)
    p6 = [t.double().requires_grad_(True) for t in (x, W1, b1, g1, be1, W2, b2)]
    x6, W16, b16, g16, be16, W26, b26 = p6
    nrm = F.group_norm((x6 @ W16.t() + b16).transpose(1, 2), 16, g16, be16, 1e-5)      # (B,C1,P)
    z6 = nrm.max(dim=2)[0]
    t6 = torch.sigmoid(F.relu(nrm).transpose(1, 2) @ W26.t() + b26)
    ((z6 * Rz.double()).sum() + (t6 * Rt.double()).sum()).backward()
    xd = x.to(dev)
    W1d, b1d, g1d, be1d, W2d, b2d = (t.to(dev) for t in (W1, b1, g1, be1, W2, b2))
    y1 = ops.conv1x1(ops.PackedWeight(W1d), b1d, xd)
    sc, sh, mean, rstd, z = T.gn_stats_train(y1, C1, g1d, be1d, want_max=True)
    t = ops.conv1x1(ops.PackedWeight(W2d), b2d, y1, in_scale=sc, in_shift=sh, in_relu=True, act=1)
    rel("head_z", z, z6, 2e-5)
    rel("head_t", t[:, :, :C2], t6, 2e-5)
    amax = T.argmax_points(y1, C1, sc, sh)
    assert torch.equal(amax.cpu().long(), nrm.max(dim=2)[1])
    dt = Rt.to(dev) * t[:, :, :C2] * (1 - t[:, :, :C2])
    dt = dt.contiguous()
    dW2, db2 = torch.empty_like(W2d), torch.empty_like(b2d)
    T.conv1x1_wgrad(dt, y1, C1, C2, dW2, db2, in_scale=sc, in_shift=sh, in_relu=True)
    da = ops.conv1x1(ops.PackedWeight(W2d.t().contiguous()), None, dt)
    dg, dbe = torch.empty_like(g1d), torch.empty_like(be1d)
    T.gn_bwd(y1, da, C1, mean, rstd, g1d, be1d, dg, dbe, relu=True, dmax=Rz.to(dev), amax=amax)
    dW1, db1 = torch.empty_like(W1d), torch.empty_like(b1d)
    T.conv1x1_wgrad(da, xd, Cin, C1, dW1, db1)
    rel("head_dW2", dW2, W26.grad, 2e-5)
    rel("head_dgamma", dg, g16.grad, 5e-5)
    rel("head_dbeta", dbe, be16.grad, 5e-5)
    rel("head_dW1", dW1, W16.grad, 5e-5)
    rel("head_db1", db1, b16.grad, 5e-4)
    cs = T.colsum_batched(da, C1)
    rel("colsum_batched", cs, da.sum(1), 1e-5)
    # max-only consumer (global PointNet, pointnet.py:41-42): dA = None
    p6b = [t.double().requires_grad_(True) for t in (x, W1, b1, g1, be1)]
    nrm_b = F.group_norm((p6b[0] @ p6b[1].t() + p6b[2]).transpose(1, 2), 16, p6b[3], p6b[4], 1e-5)
    (nrm_b.max(dim=2)[0] * Rz.double()).sum().backward()
    dy = torch.empty(B, P, C1, device=dev)
    T.gn_bwd(y1, None, C1, mean, rstd, g1d, be1d, dg, dbe, relu=False, dmax=Rz.to(dev), amax=amax, out=dy)
    T.conv1x1_wgrad(dy, xd, Cin, C1, dW1, db1)
    rel("maxonly_dW1", dW1, p6b[1].grad, 5e-5)
    rel("maxonly_dgamma", dg, p6b[3].grad, 5e-5)


def _f64_state(sd):
    out = {}
    for k, v in sd.items():
        out[k] = v.double().requires_grad_(True) if v.is_floating_point() else v
    return out


@pytest.mark.parametrize("B,T,N", [(1, 2, 1024), (2, 2, 1024)])
def test_encoder_backward_matches_oracle_autograd(B, T, N):
    """Whole TPointNet++ encoder: loss = 100*mean|tnocs - gt| + <z0, R>  (the first term is the reference's T-NOCS
    training loss, train_utils.py:160-165; the second exercises the z0 / max-pool branch).  Gradients of all 188
    encoder parameters from the HIP backward vs f64 autograd through the CPU oracle on the same weights and input."""
    from oracle import model as O
    from caspr_amd.models import CaSPR
    from caspr_amd.utils.synthetic import seeded_state_dict, dense_sequences
    dev = "cuda:0"
    m = CaSPR(pretrain_tnocs=True)
    sd = {k: v for k, v in seeded_state_dict(CaSPR().state_dict(), seed=7).items() if k.startswith("encoder.")}
    m.load_state_dict(sd)
    m = m.to(dev).train()
    x, sp = dense_sequences(B, T, N)
    R = rnd(3, B, 1600, scale=0.05)
    # oracle, f64 and f32 autograd
    def oracle_grads(dt):
        s_ = {k: (v.detach().clone().to(dt).requires_grad_(True) if v.is_floating_point() else v) for k, v in sd.items()}
        z_, t_ = O.encode(s_, x.to(dt))
        l_ = 100.0 * (t_ - sp.to(dt)).abs().mean() + (z_ * R.to(dt)).sum()
        l_.backward()
        return s_, l_.detach()
    sd6, loss6 = oracle_grads(torch.float64)
    sd3, _ = oracle_grads(torch.float32)
    # HIP
    z0, tn = m.encoder(x.to(dev))
    assert z0.requires_grad and tn.requires_grad
    loss = 100.0 * (tn - sp.to(dev)).abs().mean() + (z0 * R.to(dev)).sum()
    loss.backward()
    tag = "[%d,%d,%d]" % (B, T, N)
    rel("enc_loss" + tag, loss.detach().reshape(1), loss6.reshape(1), 1e-5)
    # The f32 gradient of this network is not a continuous function of rounding: every ReLU mask and max-pool selection
    # within rounding of a tie flips (a 1e-7 fraction of a few million elements per layer = O(1) flips per layer, each
    # worth ~1/sqrt(#elements) = 5e-4 of that layer's gradient in the L2 norm, then carried through the ~20 layers below
    # it), and GroupNorm over 16-sample neighbourhoods puts 1/sigma into the backward pass twice.  f32 autograd on the CPU
    # is itself 2e-4 (head) to 2e-2 (set abstraction) away from f64 autograd in the relative L2 norm, with the same
    # layer-by-layer growth.  Criteria:
    #   * exact where no selection is upstream: the loss value and conv3's gradient (first in the backward chain);
    #   * wiring: per parameter the MEDIAN elementwise error stays below 1e-2 of the largest entry and the L2 error below
    #     0.01 + 1.5x the f32 oracle's -- a mis-wired block (wrong operand, missing term, transposed index) is O(1) in both;
    #   * accuracy class: the whole-gradient L2 error is below 0.01 and within 1.5x of the f32 oracle's own error against f64.
    e_gpu, e_ref, n, num_g, num_r, den, bad = [], [], 0, 0.0, 0.0, 0.0, []
    for name, p in m.named_parameters():
        want = sd6[name].grad
        assert p.grad is not None, "no gradient for %s" % name
        nrm = float(want.norm())
        mx = float(want.abs().max())
        if name.endswith(".bias"):   # a conv bias in front of a one-channel-per-group GroupNorm has a zero gradient
            gw = sd6[name[:-4] + "weight"].grad
            nrm = max(nrm, float(gw.norm()) / (np.sqrt(gw.shape[1]) if gw.dim() > 1 else 1.0))
            mx = max(mx, float(gw.abs().max()))
        nrm, mx = max(nrm, 1e-9), max(mx, 1e-9)
        diff = p.grad.detach().cpu().double() - want
        dg = float(diff.norm())
        dr = float((sd3[name].grad.double() - want).norm())
        med = float(diff.abs().median()) / mx
        REPORT["enc_grad" + tag + ":" + name] = {"hip_vs_f64_l2": dg / nrm, "oracle32_vs_f64_l2": dr / nrm, "median_elem_err": med, "ref_l2": nrm}
        e_gpu.append(dg / nrm)
        e_ref.append(dr / nrm)
        num_g += dg * dg
        num_r += dr * dr
        den += float(want.norm()) ** 2
        n += 1
        assert np.isfinite(dg)
        if not (med <= 1.0e-2 and dg / nrm <= 0.01 + 1.5 * dr / nrm):
            bad.append("%s: median elementwise err %.3e (of max), L2 err %.3e" % (name, med, dg / nrm))
    # per set-abstraction LEVEL (the tensors with the largest errors; a single tensor's error there is selection noise -- it moves by
    # 2x between inputs -- but a level's aggregate is stable): the HIP backward's L2 error over all tensors of the level stays within
    # 1.0x the f32 CPU autograd's own + 6e-3 (round-3 review: the per-tensor bound alone would not notice a 1.5x regression there)
    for lvl in range(5):
        keys = [k for k in REPORT if k.startswith("enc_grad" + tag + ":") and ("set_abstractions.%d." % lvl) in k]
        g_ = np.sqrt(sum((REPORT[k]["hip_vs_f64_l2"] * REPORT[k]["ref_l2"]) ** 2 for k in keys))
        o_ = np.sqrt(sum((REPORT[k]["oracle32_vs_f64_l2"] * REPORT[k]["ref_l2"]) ** 2 for k in keys))
        n_ = np.sqrt(sum(REPORT[k]["ref_l2"] ** 2 for k in keys))
        REPORT["enc_grad_sa_level%d%s" % (lvl, tag)] = {"hip_l2": g_ / n_, "oracle32_l2": o_ / n_, "tensors": len(keys)}
        if not g_ / n_ <= 1.0 * o_ / n_ + 6e-3:
            bad.append("set abstraction level %d: block L2 err %.3e vs the f32 oracle's %.3e" % (lvl, g_ / n_, o_ / n_))
    rel("enc_flush" + tag, torch.zeros(1), torch.zeros(1), 1.0)   # writes the report file
    assert not bad, "\n".join(bad)
    assert n == 188
    tot_g, tot_r = np.sqrt(num_g / den), np.sqrt(num_r / den)
    REPORT["enc_grad_summary" + tag] = {"total_l2_hip": tot_g, "total_l2_oracle32": tot_r, "median_hip": float(np.median(e_gpu)),
                                        "median_oracle32": float(np.median(e_ref)), "max_hip": float(np.max(e_gpu)),
                                        "max_oracle32": float(np.max(e_ref))}
    rel("enc_grad_conv3" + tag, m.encoder.conv3.weight.grad, sd6["encoder.conv3.weight"].grad, 1e-4)
    # Round 1 measured HIP 2.3e-2 against the f32 oracle's 6.8e-3 and called the difference selection noise.  The round-2
    # error budget (tests/test_error_budget.py) found the cause: the first set-abstraction level's forward error (a large
    # constant in front of a per-neighbourhood GroupNorm) flipped ReLU / max selections all the way up the network.  With
    # that level's 16-channel scale on centred rows (train/encoder_grad.py, csrc/backward_points.hip) the whole-gradient
    # error is 2.7e-3 .. 3.3e-3: HALF to a QUARTER of the f32 CPU autograd's own distance from f64 (6.9e-3 .. 1.1e-2).
    # Bound: no worse than 1.5x the f32 oracle's error and below 1e-2 in absolute terms.
    assert tot_g <= 0.01 and tot_g <= 1.5 * tot_r + 1e-5, "whole-gradient L2 error %.3e vs the f32 oracle's %.3e" % (tot_g, tot_r)


def test_pretrain_step_matches_reference_golden(golden, seeded_sd):
    """SURVEY.md 8a row 19, T-NOCS pre-training variant: one `run_one_epoch` step of the REAL reference (imported with
    shimmed third-party ops, tests/golden/gen_golden.py section 5): loss = 100*mean(L1 tnocs) (train_utils.py:160-165),
    backward, Adam(lr 1e-4, 0.9/0.999, eps 1e-8).  Same weights (seed), same input."""
    from caspr_amd.models import CaSPR
    dev = "cuda:0"
    m = CaSPR(pretrain_tnocs=True)
    m.load_state_dict({k: v for k, v in seeded_sd.items() if k.startswith("encoder.")})
    m = m.to(dev).train()
    x, sp = torch.from_numpy(golden["train_x"]).to(dev), torch.from_numpy(golden["train_sp"]).to(dev)
    opt = torch.optim.Adam(m.parameters(), lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0)
    opt.zero_grad()
    losses = m(x, sp)
    assert len(losses) == 1 and losses[0].shape == (1, 2, 1024, 4)
    loss = 100.0 * losses[0][:, :, :, :4].mean()
    loss.backward()
    params = dict(m.named_parameters())
    watch = [k[len("train_pre_grad:"):] for k in golden.files if k.startswith("train_pre_grad:")]
    assert "encoder.conv3.weight" in watch
    before = {k: params[k].detach().clone() for k in watch}
    opt.step()
    rel("pretrain_loss", loss.detach().reshape(1), torch.tensor([float(golden["train_pre_loss"])]), 1e-5)
    for k in watch:
        want = torch.from_numpy(golden["train_pre_grad:" + k])
        got = params[k].grad.detach().cpu()
        l2 = float((got - want).norm() / want.norm())
        REPORT["pretrain_grad_l2:" + k] = {"rel_l2": l2}
        # conv3 sits first in the backward chain (no ReLU / max selection upstream); the others carry the f32 selection
        # noise discussed in test_encoder_backward_matches_oracle_autograd (the reference's own f32 gradient has it too)
        assert l2 <= (1e-4 if k.startswith("encoder.conv3") else 3e-2), "%s: rel L2 %.3e" % (k, l2)
        # Adam's first step is -lr * g / (|g| + eps): compare where the reference gradient is clear of eps
        d_want = torch.from_numpy(golden["train_pre_delta:" + k])
        d_got = (params[k].detach() - before[k]).cpu()
        clear = want.abs() > 1e-5
        frac_bad = float(((d_got - d_want).abs() > 2e-6)[clear].float().mean())
        REPORT["pretrain_delta_bad_frac:" + k] = {"frac": frac_bad}
        assert frac_bad <= (0.0 if k.startswith("encoder.conv3") else 0.01), "%s: %.4f of the Adam deltas differ" % (k, frac_bad)
    rel("pretrain_grad_conv3", params["encoder.conv3.weight"].grad, torch.from_numpy(golden["train_pre_grad:encoder.conv3.weight"]), 1e-4)


def test_full_training_step_matches_reference_golden(golden, seeded_sd):
    """one full `run_one_epoch` step of the REAL reference (shimmed third-party ops, RK4 in place of
    dopri5 with the same step counts, fixed Hutchinson noise): loss = 0.01*mean_{b,t}(sum_n nll) + 100*mean(L1 tnocs)
    (train_utils.py:151-165), backward through CNF, latent ODE and encoder, Adam(lr 1e-4).  Checked: both returned
    tensors, the scalar loss, gradients along the whole backward chain, the Adam updates of `encoder.conv3.weight` and
    `point_cnf.chain.1.sqrt_end_time`, and the MovingBatchNorm running statistics after the step."""
    from caspr_amd.models import CaSPR
    dev = "cuda:0"
    m = CaSPR(cnf_rk4_steps=8, latent_rk4_steps=4)
    m.load_state_dict(seeded_sd)
    m = m.to(dev).train()
    x, sp = torch.from_numpy(golden["train_x"]).to(dev), torch.from_numpy(golden["train_sp"]).to(dev)
    e = torch.from_numpy(golden["train_e"]).to(dev)
    opt = torch.optim.Adam(m.parameters(), lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0)
    opt.zero_grad()
    nll, tl = m(x, sp, e=e)
    rel("train_full_nll", nll, torch.from_numpy(golden["train_full_nll"]), 2e-5)
    rel("train_full_tnocs_l1", tl, torch.from_numpy(golden["train_full_tnocs_l1"]), 2e-5)
    loss = 0.01 * nll.sum(2).mean() + 100.0 * tl[:, :, :, :4].mean()
    loss.backward()
    rel("train_full_loss", loss.detach().reshape(1), torch.tensor([float(golden["train_full_loss"])]), 1e-5)
    params = dict(m.named_parameters())
    watch = [k[len("train_full_grad:"):] for k in golden.files if k.startswith("train_full_grad:")]
    assert "point_cnf.chain.1.sqrt_end_time" in watch and "latent_ode.ode_func.dynamics_net.0.weight" in watch
    before = {k: params[k].detach().clone() for k in watch}
    opt.step()
    exact = ("encoder.conv3", "point_cnf.")      # no ReLU / max selection upstream of these in the backward chain
    bad_grads = []
    for k in watch:
        want = torch.from_numpy(golden["train_full_grad:" + k])
        got = params[k].grad.detach().cpu()
        l2 = float((got - want).norm() / want.norm())
        REPORT["train_full_grad_l2:" + k] = {"rel_l2": l2}
        tol = 2e-3 if k.startswith(exact) else 3e-2
        if not l2 <= tol:
            bad_grads.append("%s: rel L2 %.3e > %.1e" % (k, l2, tol))
    rel("train_full_flush", torch.zeros(1), torch.zeros(1), 1.0)
    assert not bad_grads, "\n".join(bad_grads)
    for k in ("encoder.conv3.weight", "point_cnf.chain.1.sqrt_end_time"):
        d_want = torch.from_numpy(golden["train_full_delta:" + k])
        d_got = (params[k].detach() - before[k]).cpu()
        clear = torch.from_numpy(golden["train_full_grad:" + k]).abs() > 1e-5
        bad = float(((d_got - d_want).abs() > 2e-6)[clear].float().mean()) if bool(clear.any()) else 0.0
        REPORT["train_full_delta_bad_frac:" + k] = {"frac": bad}
        assert bad == 0.0, "%s: Adam update differs" % k
    rel("train_full_mbn_running_mean", m.point_cnf.chain[0].running_mean, torch.from_numpy(golden["train_full_mbn_running_mean"]), 1e-5)
    rel("train_full_mbn_running_var", m.point_cnf.chain[0].running_var, torch.from_numpy(golden["train_full_mbn_running_var"]), 1e-5)


def test_train_loop_checkpoint_resume(tmp_path, seeded_sd):
    """train.py:135-190 cadence on the HIP path: two epochs of T-NOCS pre-training over a two-batch loader, periodic and
    BEST checkpoints in the reference's state_dict format, and a resumed run that continues bit-identically."""
    from caspr_amd.models import CaSPR
    from caspr_amd.train.loop import train
    from caspr_amd.utils.synthetic import dense_sequences
    dev = torch.device("cuda:0")
    enc_sd = {k: v for k, v in seeded_sd.items() if k.startswith("encoder.")}
    batches = [[dense_sequences(1, 2, 1024, seed=s)] for s in (1, 2)]
    logs = []

    def fresh():
        m = CaSPR(pretrain_tnocs=True)
        m.load_state_dict(enc_sd)
        return m.to(dev)
    m = fresh()
    val = train(m, batches, batches[:1], dev, str(tmp_path), num_epochs=2, log=logs.append)
    assert len(val) == 2 and val[1] < val[0], "validation loss did not go down: %s" % val
    for f in ("time_model_0.pth", "time_model_1.pth", "BEST_time_model.pth", "resume_0.pth"):
        assert os.path.exists(os.path.join(str(tmp_path), f)), f
    saved = torch.load(os.path.join(str(tmp_path), "time_model_1.pth"), map_location="cpu")
    assert list(saved.keys()) == list(CaSPR(pretrain_tnocs=True).state_dict().keys())
    # resume from the end of epoch 0 and redo epoch 1: same weights as the uninterrupted run (deterministic reductions;
    # the float-atomic scatter-adds may move the last bit)
    m2 = fresh()
    out2 = tmp_path / "resumed"
    out2.mkdir()
    train(m2, batches, batches[:1], dev, str(out2), num_epochs=2, log=logs.append, resume=os.path.join(str(tmp_path), "resume_0.pth"))
    # Same weights as the uninterrupted run.  Not bit-identical: the two float-atomic scatter-adds move the last bit of some
    # gradients, and Adam's first steps (update = m / (sqrt(v) + 1e-8)) turn that into a few 1e-6 on the entries whose
    # gradient is itself tiny (observed: up to 5e-6 on 2-5 of the 144 entries of the first set-abstraction conv).  A resume
    # that lost the optimizer state or the epoch would move EVERY entry by ~lr = 1e-4: require max <= 2e-5, median <= 1e-6.
    for key in ("encoder.conv3.weight", "encoder.local_extract.set_abstractions.0.pointnet_modules.0.conv_layers.0.weight",
                "encoder.conv1.weight"):
        a, b = m.state_dict()[key], m2.state_dict()[key]
        d = (a - b).abs()
        REPORT["resume:" + key] = {"median_abs_diff": float(d.median()), "max_abs_diff": float(d.max())}
        assert float(d.max()) <= 2e-5 and float(d.median()) <= 1e-6, "%s: max %.3e median %.3e" % (key, float(d.max()), float(d.median()))
    rel("resume_flush", torch.zeros(1), torch.zeros(1), 1.0)


def test_train_checkpoint_reconstruct_end_to_end(tmp_path):
    """The whole surface in one go (tools/train_and_eval.py, which profiles/ keeps a 300-step and a 2000-step record of): 40 full training
    steps on fresh synthetic car sequences (the reference's loss and Adam settings), the checkpoint in the reference's format, loaded back
    the way test.py:104-107 / `bench.py --weights` load one, and then on the TRAINED weights: better held-out Chamfer and T-NOCS error than
    before training, the run-time accuracy guard quiet at the default step counts, and reconstruct() within 1e-5 of the f64 oracle."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("train_and_eval", os.path.join(ROOT, "tools", "train_and_eval.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    rep = mod.main(["--steps", "40", "--eval-seqs", "1", "--ckpt", str(tmp_path / "time_model_0.pth"), "--no-dopri5", "--no-headline"])
    curve = rep["train"]["curve"]
    assert rep["train"]["finite"]
    assert curve[-1]["loss"] < 0.5 * curve[0]["loss"], curve
    assert rep["checkpoint"]["round_trip_bitwise"] and rep["checkpoint"]["keys"] == 238
    before, after = rep["held_out_before"], rep["held_out_after"]
    assert after["chamfer_x1000"]["mean"] < 0.5 * before["chamfer_x1000"]["mean"], (before, after)
    assert after["tnocs_space_l2"]["mean"] < before["tnocs_space_l2"]["mean"], (before, after)
    assert rep["guard_at_8_and_2_steps"]["verdict"] == "quiet", rep["guard_at_8_and_2_steps"]
    par = rep["parity_trained_weights"]["hip_vs_f64_oracle_same_rk4_map"]
    REPORT["end_to_end_trained_weights"] = {"loss_first": curve[0]["loss"], "loss_last": curve[-1]["loss"], "chamfer_x1000_before": before["chamfer_x1000"]["mean"],
                                            "chamfer_x1000_after": after["chamfer_x1000"]["mean"], "hip_vs_f64_x": par["x"], "hip_vs_f64_tnocs": par["tnocs"]}
    rel("end_to_end_flush", torch.zeros(1), torch.zeros(1), 1.0)
    assert par["ok"], par


@pytest.mark.parametrize("kw", [dict(regress_tnocs=False), dict(cnf_blocks=2), dict(augment_quad=False, augment_pairs=False)])
def test_training_variants_step(kw, seeded_sd):
    """Constructor variants the reference configs use (cfg-4: regress_tnocs=False; cnf_blocks; no input augmentation):
    three optimisation steps run and every trainable parameter the loss depends on receives a finite gradient.  (The
    loss itself need not fall monotonically over three steps: the MovingBatchNorm running statistics that normalise the
    CNF's input and output move by 10% towards the batch statistics at every training-mode call.)"""
    from caspr_amd.models import CaSPR
    from caspr_amd.train.loop import train_step
    from caspr_amd.utils.synthetic import dense_sequences, seeded_state_dict
    dev = torch.device("cuda:0")
    m = CaSPR(cnf_rk4_steps=4, latent_rk4_steps=2, **kw)
    m.load_state_dict(seeded_state_dict(m.state_dict(), 3))
    m = m.to(dev)
    x, sp = (t.to(dev) for t in dense_sequences(1, 2, 1024, seed=11))
    e = rnd(5, 2, 1024, 3).to(dev)
    opt = torch.optim.Adam(m.parameters(), lr=1e-4)
    losses = [train_step(m, opt, x, sp, e=e)[0] for _ in range(3)]
    assert all(np.isfinite(losses)), losses
    missing = [n for n, p in m.named_parameters() if p.grad is None and not (kw.get("regress_tnocs") is False and n.startswith("encoder.conv3"))]
    assert not missing, "no gradient for %s" % missing[:5]
    assert all(bool(torch.isfinite(p.grad).all()) for p in m.parameters() if p.grad is not None)


def test_full_step_all_flow_parameters_vs_f64_oracle(golden, seeded_sd):
    """Every latent-ODE / CNF / MovingBatchNorm parameter (51 tensors: all four gated layers with their hyper networks,
    the dynamics MLP, sqrt_end_time): HIP training-step gradients against f64 autograd through the oracle's
    differentiable mode (itself pinned to the real reference's gradients in tests/test_oracle_golden.py)."""
    from oracle import model as O
    from caspr_amd.models import CaSPR
    dev = "cuda:0"
    skip = ("running_mean", "running_var", "step", "_num_evals")
    sd6 = {k: (v.detach().clone().double().requires_grad_(True) if v.is_floating_point() and not k.endswith(skip) else
               (v.double() if v.is_floating_point() else v)) for k, v in seeded_sd.items()}
    x, sp, e = (torch.from_numpy(golden[k]) for k in ("train_x", "train_sp", "train_e"))
    loss6, _, _ = O.training_loss(sd6, x.double(), sp.double(), e.double(), cnf_steps=8, latent_steps=4)
    loss6.backward()
    m = CaSPR(cnf_rk4_steps=8, latent_rk4_steps=4)
    m.load_state_dict(seeded_sd)
    m = m.to(dev).train()
    nll, tl = m(x.to(dev), sp.to(dev), e=e.to(dev))
    loss = 0.01 * nll.sum(2).mean() + 100.0 * tl[:, :, :, :4].mean()
    loss.backward()
    rel("full64_loss", loss.detach().reshape(1), loss6.detach().reshape(1), 1e-5)
    n, bad = 0, []
    for name, p in m.named_parameters():
        if name.startswith("encoder."):
            continue        # encoder gradients: test_encoder_backward_matches_oracle_autograd (selection noise discussed there)
        key = name.replace("latent_ode.solver.ode_func", "latent_ode.ode_func")
        want = sd6[key].grad
        assert want is not None and p.grad is not None, name
        err = float((p.grad.detach().cpu().double() - want).norm() / want.norm().clamp_min(1e-12))
        REPORT["full64_grad_l2:" + name] = {"rel_l2": err, "ref_l2": float(want.norm())}
        n += 1
        if not err <= 2e-4:
            bad.append("%s: rel L2 %.3e" % (name, err))
    rel("full64_flush", torch.zeros(1), torch.zeros(1), 1.0)
    assert n >= 30, n
    assert not bad, "\n".join(bad)


def test_sharded_gradient_equals_batch_gradient(seeded_sd):
    """SURVEY.md 8e on the real model: the gradient of the training loss over a 2-sequence batch equals the average of
    the two single-sequence gradients (what two ranks + GradBucket.all_reduce_mean produce) -- every stage is
    per-sequence (GroupNorm per sample, fixed-step integrators, MovingBatchNorm using the pre-update statistics)."""
    from caspr_amd.models import CaSPR
    from caspr_amd.train.loop import training_loss
    from caspr_amd.utils.synthetic import dense_sequences
    dev = torch.device("cuda:0")
    x, sp = (t.to(dev) for t in dense_sequences(2, 2, 1024, seed=61))
    e = rnd(7, 4, 1024, 3).to(dev)

    def grads(xs, sps, es):
        m = CaSPR(cnf_rk4_steps=4, latent_rk4_steps=2)
        m.load_state_dict(seeded_sd)
        m = m.to(dev).train()
        loss, _, _ = training_loss(m(xs, sps, e=es), 0.01, 100.0)
        loss.backward()
        return {n: p.grad.detach().clone() for n, p in m.named_parameters()}, float(loss.detach())
    g_all, l_all = grads(x, sp, e)
    g0, l0 = grads(x[:1], sp[:1], e[:2])
    g1, l1 = grads(x[1:], sp[1:], e[2:])
    assert abs(l_all - 0.5 * (l0 + l1)) <= 1e-5 * abs(l_all)
    num = den = 0.0
    worst = ("", 0.0)
    for n in g_all:
        avg = 0.5 * (g0[n] + g1[n])
        d, r = float((g_all[n] - avg).norm()), float(avg.norm())
        num, den = num + d * d, den + r * r
        if d > worst[1]:
            worst = (n, d)
    REPORT["sharded_grad"] = {"total_rel_l2": (num / den) ** 0.5, "worst": worst[0], "worst_abs_l2": worst[1], "grad_l2": den ** 0.5}
    rel("sharded_grad_flush", torch.zeros(1), torch.zeros(1), 1.0)
    # same kernels on the same per-sequence data: only reduction order across the batch differs (weight-gradient slabs)
    # (per tensor the difference is measured against the whole gradient's norm: some biases have a mathematically zero gradient)
    assert (num / den) ** 0.5 <= 1e-5 and worst[1] <= 1e-5 * den ** 0.5, REPORT["sharded_grad"]


def test_encoder_gradients_are_bit_reproducible(seeded_sd):
    """Every reduction of the training path runs in a fixed order (slab / block combines, segment gathers instead of float
    atomics): two backward passes over the same input give bit-identical gradients."""
    from caspr_amd.models import CaSPR
    from caspr_amd.utils.synthetic import car_sequences
    dev = torch.device("cuda:0")
    m = CaSPR(pretrain_tnocs=True)
    m.load_state_dict({k: v for k, v in seeded_sd.items() if k.startswith("encoder.")})
    m = m.to(dev).train()
    x, sp = (t.to(dev) for t in car_sequences(2, 2, 1024, seed=71))
    R = rnd(9, 2, 1600, scale=0.05).to(dev)

    def grads():
        m.zero_grad()
        z0, tn = m.encoder(x)
        (100.0 * (tn - sp).abs().mean() + (z0 * R).sum()).backward()
        return [p.grad.detach().clone() for p in m.parameters()]
    a, b = grads(), grads()
    bad = [n for (n, _), u, v in zip(m.named_parameters(), a, b) if not torch.equal(u, v)]
    assert not bad, "gradients differ between two identical passes: %s" % bad[:5]


def test_full_model_gradients_are_bit_reproducible(seeded_sd, golden):
    """The same for the whole training step (encoder + latent ODE + CNF with divergence)."""
    from caspr_amd.models import CaSPR
    dev = torch.device("cuda:0")
    m = CaSPR(cnf_rk4_steps=2, latent_rk4_steps=2)
    m.load_state_dict(seeded_sd)
    m = m.to(dev).train()
    x, sp = torch.from_numpy(golden["train_x"]).to(dev), torch.from_numpy(golden["train_sp"]).to(dev)
    e = torch.from_numpy(golden["train_e"]).to(dev)
    stats = {k: v.clone() for k, v in m.state_dict().items() if "running_" in k or k.endswith(".step")}

    def grads():
        m.load_state_dict(stats, strict=False)     # MovingBatchNorm statistics move at every training-mode call: rewind them
        m.zero_grad()
        nll, tl = m(x, sp, e=e)
        (0.01 * nll.sum(2).mean() + 100.0 * tl[:, :, :, :4].mean()).backward()
        return [p.grad.detach().clone() for p in m.parameters()]
    a, b = grads(), grads()
    bad = [n for (n, _), u, v in zip(m.named_parameters(), a, b) if not torch.equal(u, v)]
    assert not bad, "gradients differ between two identical passes: %s" % bad[:5]


def test_cnf_step_checkpointing_gives_the_same_bits(seeded_sd, golden):
    """config.train_cnf_checkpoint: the CNF's tape kept per RK4 step and the step recomputed in the backward pass (the reference's
    adjoint re-integrates too, cnf.py:100-110).  A step is a pure function of its inputs on deterministic kernels: loss and every
    gradient are bit-identical to the taped form, and the peak memory of the step is smaller."""
    from caspr_amd.models import CaSPR
    from caspr_amd.train import flow_grad
    dev = torch.device("cuda:0")
    m = CaSPR(cnf_rk4_steps=4, latent_rk4_steps=2)
    m.load_state_dict(seeded_sd)
    m = m.to(dev).train()
    x, sp = torch.from_numpy(golden["train_x"]).to(dev), torch.from_numpy(golden["train_sp"]).to(dev)
    e = torch.from_numpy(golden["train_e"]).to(dev)
    stats = {k: v.clone() for k, v in m.state_dict().items() if "running_" in k or k.endswith(".step")}

    def grads(ck):
        prev, flow_grad.CHECKPOINT_STEPS = flow_grad.CHECKPOINT_STEPS, ck
        try:
            m.load_state_dict(stats, strict=False)
            m.zero_grad()
            torch.cuda.synchronize()
            torch.cuda.reset_peak_memory_stats()
            nll, tl = m(x, sp, e=e)
            loss = 0.01 * nll.sum(2).mean() + 100.0 * tl[:, :, :, :4].mean()
            loss.backward()
            torch.cuda.synchronize()
            return float(loss), [None if p.grad is None else p.grad.detach().clone() for p in m.parameters()], torch.cuda.max_memory_allocated()
        finally:
            flow_grad.CHECKPOINT_STEPS = prev
    la, ga, ma = grads(False)
    lb, gb, mb = grads(True)
    assert la == lb
    bad = [n for (n, _), u, v in zip(m.named_parameters(), ga, gb) if (u is None) != (v is None) or (u is not None and not torch.equal(u, v))]
    assert not bad, "checkpointed and taped gradients differ: %s" % bad[:5]
    REPORT["cnf_step_checkpoint_peak_bytes"] = {"taped": ma, "checkpointed": mb}
    assert mb < ma


def test_eval_mode_forward_is_differentiable_and_cnf_forward_draws_fresh_noise(seeded_sd, golden):
    """The reference's forward stays differentiable in eval() (its callers add torch.no_grad() themselves): same loss values
    as under no_grad, a graph when grad mode is on, no MovingBatchNorm statistics update.  CNF.forward (cnf.py:100) clears
    the Hutchinson noise of the previous solve instead of re-installing it; ops.cnf_rk4 rejects mismatched e / logp."""
    from caspr_amd import ops
    from caspr_amd.models import CaSPR
    dev = "cuda:0"
    m = CaSPR(cnf_rk4_steps=2, latent_rk4_steps=1)
    m.load_state_dict(seeded_sd)
    m = m.to(dev).eval()
    x, sp = torch.from_numpy(golden["train_x"]).to(dev), torch.from_numpy(golden["train_sp"]).to(dev)     # dense (1,2,1024,4): well-conditioned
    e = torch.from_numpy(golden["train_e"]).to(dev)
    rm0 = m.point_cnf.chain[0].running_mean.clone()
    with torch.no_grad():
        nll0, tl0 = m(x, sp, e=e)
    assert not nll0.requires_grad
    nll, tl = m(x, sp, e=e)
    assert nll.requires_grad and tl.requires_grad
    rel("evalgrad_nll", nll, nll0, 5e-5)        # taped path (materialised neighbourhoods, unfused layers) vs the fused inference kernels
    rel("evalgrad_tnocs", tl, tl0, 5e-5)
    (0.01 * nll.sum(2).mean() + 100.0 * tl[:, :, :, :4].mean()).backward()
    g = m.encoder.conv3.weight.grad
    assert g is not None and bool(torch.isfinite(g).all()) and float(g.abs().max()) > 0
    assert m.point_cnf.chain[1].sqrt_end_time.grad is not None
    assert torch.equal(m.point_cnf.chain[0].running_mean, rm0)          # eval(): frozen statistics
    # CNF.forward: the noise of the previous solve must not be reused (and a larger x must not read past it)
    cnf = m.point_cnf.chain[1]
    ctx = torch.randn(2, m.cnf_args.zdim, device=dev)
    with torch.no_grad():
        cnf.odefunc.before_odeint(torch.randn(2, 8, 3, device=dev))
        xb = torch.randn(2, 64, 3, device=dev)
        y1, lp1 = cnf(xb, ctx, torch.zeros(2, 64, 1, device=dev))
        e1 = cnf.odefunc._e
        assert tuple(e1.shape) == (2, 64, 3)
        y2, lp2 = cnf(xb, ctx, torch.zeros(2, 64, 1, device=dev))
        assert not torch.equal(cnf.odefunc._e, e1) and torch.equal(y1, y2) and not torch.equal(lp1, lp2)
        w = cnf._weights()
        hyper = ops.conv1x1(w["hyp"], w["hyp_bias"], ctx.view(1, 2, -1))[0]
        with pytest.raises(ValueError):
            ops.cnf_rk4(xb, hyper, w["tcol"], w["w0"], w["b0"], w["w1p"], w["b1"], w["w2p"], w["b2"], w["w3"], w["b3"], 0.5, 2, False,
                        e=torch.randn(2, 8, 3, device=dev), logp=torch.zeros(2, 64, 1, device=dev))


@pytest.mark.parametrize("team,nseq", [(True, 8), (False, 8), (True, 20)])
def test_latent_solve_single_node_matches_per_layer_autograd_and_f64(monkeypatch, team, nseq):
    """team: the one-launch forms (caspr_latent_rk4_team_tape_f32 / _adjoint_f32: 32 workgroups per 16 sequences, the tape and the deltas
    written by the kernels) against the launch-per-product form of the same node; 20 sequences = two teams, the second one ragged.
    LatentSolve (one autograd node, hand-written reverse sweep, weight gradients of all evaluations as one product per layer)
    against the per-layer form differentiated by torch.autograd on the same kernels, and against float64 autograd of the same
    RK4 map on the CPU (latent_ode_model.py:45-70,139-147 with the fixed-step solver of DESIGN.md section 4)."""
    from caspr_amd.models.latent_ode_model import LatentODE
    from caspr_amd.train import flow_grad as FG
    dev = torch.device("cuda:0")
    torch.manual_seed(11)
    lat = LatentODE(input_size=64, hidden_size=512, num_layers=2).to(dev)
    lat.rk4_steps = 2
    monkeypatch.setattr(FG.LatentSolve, "TEAM", team)
    z0 = rnd(1, nseq, 64, scale=0.5).to(dev).requires_grad_(True)
    times = torch.tensor([0.0, 0.2, 0.2, 0.55, 1.0], device=dev)          # a repeated stamp: a zero-length interval
    wgt = rnd(2, nseq, 5, 64).to(dev)
    params = [p for p in lat.ode_func.parameters()]

    def grads(fn):
        for p in params:
            p.grad = None
        z0.grad = None
        out = fn(lat, z0, times)
        (out * wgt).sum().backward()
        return out.detach(), z0.grad.clone(), [p.grad.clone() for p in params]
    out_n, gz_n, gp_n = grads(lambda l, z, t: FG.LatentSolve.apply(z, t, l.rk4_steps, *[x for i in (0, 2, 4, 6) for x in (l.ode_func.dynamics_net[i].weight, l.ode_func.dynamics_net[i].bias)]))
    out_l, gz_l, gp_l = grads(FG.latent_solve_layers)
    rel("latent_node_vs_layers_out", out_n, out_l, 1e-6)
    rel("latent_node_vs_layers_dz0", gz_n, gz_l, 2e-5)
    for i, (a, b) in enumerate(zip(gp_n, gp_l)):
        rel("latent_node_vs_layers_dp%d" % i, a, b, 2e-5)
    # float64 autograd of the same map
    lin = [lat.ode_func.dynamics_net[i] for i in (0, 2, 4, 6)]
    W = [l.weight.detach().cpu().double().requires_grad_(True) for l in lin]
    Bi = [l.bias.detach().cpu().double().requires_grad_(True) for l in lin]
    z64 = z0.detach().cpu().double().requires_grad_(True)

    def f(z):
        h = z
        for i in range(4):
            h = h @ W[i].t() + Bi[i]
            if i < 3:
                h = torch.tanh(h)
        return h
    outs, z, tt = [z64], z64, times.cpu().double()
    for k in range(1, tt.shape[0]):
        h = (tt[k] - tt[k - 1]) / 2
        for _ in range(2):
            k1 = f(z); k2 = f(z + 0.5 * h * k1); k3 = f(z + 0.5 * h * k2); k4 = f(z + h * k3)
            z = z + (h / 6.0) * (k1 + 2.0 * k2 + 2.0 * k3 + k4)
        outs.append(z)
    out64 = torch.stack(outs, dim=1)
    (out64 * wgt.cpu().double()).sum().backward()
    rel("latent_node_vs_f64_out", out_n, out64, 2e-6)
    rel("latent_node_vs_f64_dz0", gz_n, z64.grad, 3e-5)
    want = [g for pair in zip([w.grad for w in W], [b.grad for b in Bi]) for g in pair]
    for i, (a, b) in enumerate(zip(gp_n, want)):
        rel("latent_node_vs_f64_dp%d" % i, a, b, 3e-5)


def test_latent_solve_with_one_time_stamp_is_the_identity():
    """One time stamp: the solve returns its initial state (odeint at a single time, latent_ode_model.py:58-66), dL/dz0 is the output's
    gradient and the dynamics net gets zero gradients -- not an exception from an empty tape."""
    from caspr_amd.models.latent_ode_model import LatentODE
    from caspr_amd.train import flow_grad as FG
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    lat = LatentODE(input_size=64, hidden_size=512, num_layers=2).to(dev)
    z0 = rnd(1, 4, 64).to(dev).requires_grad_(True)
    wgt = rnd(2, 4, 1, 64).to(dev)
    wb = [x for i in (0, 2, 4, 6) for x in (lat.ode_func.dynamics_net[i].weight, lat.ode_func.dynamics_net[i].bias)]
    out = FG.LatentSolve.apply(z0, torch.tensor([0.3], device=dev), 2, *wb)
    assert out.shape == (4, 1, 64) and torch.equal(out[:, 0], z0.detach())
    (out * wgt).sum().backward()
    assert torch.equal(z0.grad, wgt[:, 0])
    assert all(p.grad is not None and float(p.grad.abs().max()) == 0.0 for p in wb)


def test_cnf_fused_layers_match_the_separate_passes():
    """The ODE function's hidden layers with the gated softplus in the conv's epilogue (CnfLayer / CnfLayerOut, row layout blk = 32)
    against the separate passes (linear_rows + CnfAct in the [values | tangents] layout, blk = R), forward and every gradient, and
    the first layer (CnfIn) in both layouts: same values up to the rounding of the hardware transcendentals."""
    from caspr_amd.train import flow_grad as FG
    dev = torch.device("cuda:0")
    BT, n, C = 3, 128, 512
    R = BT * n
    y, e = rnd(1, R, 3).to(dev).requires_grad_(True), rnd(2, R, 3).to(dev)
    w0, b0 = rnd(3, C, 3, scale=0.5).to(dev).requires_grad_(True), rnd(4, C, scale=0.2).to(dev).requires_grad_(True)
    w1, b1 = rnd(5, C, C, scale=1.0 / np.sqrt(C)).to(dev).requires_grad_(True), rnd(6, C, scale=0.2).to(dev).requires_grad_(True)
    w2, b2 = rnd(7, C, C, scale=1.0 / np.sqrt(C)).to(dev).requires_grad_(True), rnd(8, C, scale=0.2).to(dev).requires_grad_(True)
    wo = rnd(9, 3, C, scale=1.0 / np.sqrt(C)).to(dev).requires_grad_(True)
    gates = [torch.sigmoid(rnd(10 + i, BT, C)).to(dev).requires_grad_(True) for i in range(3)]
    betas = [rnd(20 + i, BT, C, scale=0.3).to(dev).requires_grad_(True) for i in range(3)]
    leaves = [y, w0, b0, w1, b1, w2, b2, wo] + gates + betas
    wgt = rnd(30, 2 * R, 3).to(dev)

    def to_pts(z, blk):                     # (2R, c) rows in layout blk -> (2, R, c): values, tangents by point
        zz = z.view(R // blk, 2, blk, -1)
        return torch.stack([zz[:, 0].reshape(R, -1), zz[:, 1].reshape(R, -1)])

    def run(fused):
        for t in leaves:
            t.grad = None
        blk = 32 if fused else R
        h = FG.CnfIn.apply(y, e, w0, b0, gates[0], betas[0], n, blk)
        if fused == "node":          # both hidden layers + output product as one node (first layer's backward in the dgrad epilogue)
            zo = FG.CnfHidden.apply(h, w1, b1, gates[1], betas[1], w2, b2, gates[2], betas[2], wo, n)
        elif fused:
            h1 = FG.CnfLayer.apply(h, w1, b1, gates[1], betas[1], n)
            zo = FG.CnfLayerOut.apply(h1, w2, b2, gates[2], betas[2], wo, n)
        else:
            h1 = FG.CnfAct.apply(FG.linear_rows(h, w1, None), b1, gates[1], betas[1], n, blk)
            h2 = FG.CnfAct.apply(FG.linear_rows(h1, w2, None), b2, gates[2], betas[2], n, blk)
            zo = FG.linear_rows(h2, wo, None)
        out = to_pts(zo, blk)
        (out * wgt.view(2, R, 3)).sum().backward()
        return out.detach(), to_pts(h.detach(), blk), [t.grad.clone() for t in leaves]
    out_f, h_f, g_f = run(True)
    out_s, h_s, g_s = run(False)
    rel("cnf_fused_in_layer", h_f, h_s, 1e-6)
    rel("cnf_fused_out", out_f, out_s, 3e-6)
    names = ["y", "w0", "b0", "w1", "b1", "w2", "b2", "wo", "gate0", "gate1", "gate2", "beta0", "beta1", "beta2"]
    for nm, a, b in zip(names, g_f, g_s):
        rel("cnf_fused_grad_" + nm, a, b, 2e-5)
    out_n, h_n, g_n = run("node")
    rel("cnf_node_out", out_n, out_s, 3e-6)
    for nm, a, b in zip(names, g_n, g_s):
        rel("cnf_node_grad_" + nm, a, b, 2e-5)
