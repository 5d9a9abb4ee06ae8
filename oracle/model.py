"""oracle/model.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Functional PyTorch-CPU restatement of the reference's CaSPR encode -> advect -> sample path,
driven by a plain state_dict with the reference's key names.  Every
function cites the reference file:line it follows (paths relative to /root/reference/caspr/).
The third-party operators come from oracle/point_ops (C).  Two integrators: fixed-step RK4 (what
the HIP kernels run; the parity gate) and an adaptive Dormand-Prince restatement of
torchdiffeq 0.0.1 `dopri5` (what the reference runs; recalled from the public sources, PARITY
UNPINNED; used only to report the RK4-vs-dopri5 gap).

Pinned against the real reference's pure-torch modules by tests/golden/gen_golden.py (run in the
build container, where /root/reference can be imported behind sys.modules shims) and
tests/test_oracle_golden.py (runs anywhere, from the committed fixtures).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import math

import torch
import torch.nn.functional as F

from . import point_ops as P

NUM_GROUPS = 16  # models/pointnet2.py:12

# models/pointnet2.py:62-146 -- (num_points_out, [mlp A, mlp B]); radii from radii_list[l], [l+1]
SA_SPECS = [
    (1024, [[16, 16, 32], [32, 32, 64]]),
    (512, [[32, 32, 64], [32, 32, 64]]),
    (256, [[64, 64, 128], [64, 96, 128]]),
    (64, [[128, 256, 256], [128, 256, 256]]),
    (16, [[256, 256, 512], [256, 256, 512]]),
]
SA_NS = [16, 32]
DEFAULT_RADII = [0.02, 0.05, 0.1, 0.2, 0.4, 0.8]


def _conv(sd, key, x):
    """nn.Conv1d(k=1) on (B,C,P)."""
    return F.conv1d(x, sd[key + ".weight"], sd[key + ".bias"])


def _gn(sd, key, x):
    return F.group_norm(x, NUM_GROUPS, sd[key + ".weight"], sd[key + ".bias"], 1e-5)


# ----------------------------------------------------------------------------------------------
# encoder
# ----------------------------------------------------------------------------------------------
def pointnet_global(sd, x, pre="encoder.global_extract"):
    """models/pointnet.py:34-46.  x (B,4,P) -> (B,1024+64,P) [tiled max feature, then point feature]."""
    n_pts = x.shape[2]
    x = F.relu(_gn(sd, pre + ".bn1", _conv(sd, pre + ".conv1", x)))
    pointfeat = x
    x = F.relu(_gn(sd, pre + ".bn2", _conv(sd, pre + ".conv2", x)))
    x = _gn(sd, pre + ".bn3", _conv(sd, pre + ".conv3", x))
    g = torch.max(x, 2, keepdim=True)[0]
    return torch.cat([g.repeat(1, 1, n_pts), pointfeat], 1)


def feature_extractor(sd, pre, x):
    """models/pointnet2.py:649-703 (global_feat=True, transposed_input=True): (G,C,ns) -> (G,C3)."""
    n_layers = 3
    for l in range(n_layers):
        x = _gn(sd, "%s.bn_layers.%d" % (pre, l), _conv(sd, "%s.conv_layers.%d" % (pre, l), x))
        if l < n_layers - 1:
            x = F.relu(x)
    return torch.max(x, 2)[0]


def set_abstraction(sd, pre, xyz, feat, M, radii, intermediates=None):
    """models/pointnet2.py:361-419.  xyz (B,n,3), feat (B,C,n) -> new_xyz (B,M,3), (B,Cout,M)."""
    B = xyz.shape[0]
    idx = P.furthest_point_sampling(xyz, M)                                   # :384
    new_xyz = P.fps_gather_by_index(xyz.transpose(1, 2).contiguous(), idx)    # :385
    new_xyz = new_xyz.transpose(1, 2).contiguous()                            # :387
    outs = []
    ball_idx = []
    for i, ns in enumerate(SA_NS):
        bidx = P.ball_query(radii[i], ns, xyz, new_xyz)                       # :391 (grouper)
        ball_idx.append(bidx)
        g = P.group(xyz, new_xyz, feat, bidx)                                 # (B,M,C+3,ns)
        g = g.view(-1, g.shape[2], ns)                                        # :397
        f = feature_extractor(sd, "%s.pointnet_modules.%d" % (pre, i), g)     # :401
        outs.append(f.view(B, M, -1).transpose(1, 2))                         # :407
    if intermediates is not None:
        intermediates.append({"fps_idx": idx, "ball_idx": ball_idx, "new_xyz": new_xyz})
    return new_xyz, torch.cat(outs, dim=1).contiguous()                       # :413


def feature_propagator(sd, pre, xyz, xyz_prev, feat, feat_prev):
    """models/pointnet2.py:483-525."""
    dist, idx = P.three_nn(xyz, xyz_prev)                                     # :514
    if xyz.dtype == torch.float64:
        dist = P.three_nn_f64(xyz, xyz_prev, idx)
    inverse_dist = 1.0 / (dist + 1e-8)                                        # :516
    total = torch.sum(inverse_dist, dim=2, keepdim=True)
    weights = inverse_dist / total                                            # :518
    new = P.three_interpolate(feat_prev, idx, weights)                        # :519
    if feat is not None:
        new = torch.cat([new, feat], dim=1)                                   # :523
    for l in (0, 3):                                                          # :525 Sequential
        new = F.relu(_gn(sd, "%s.unit_pointnet.%d" % (pre, l + 1), _conv(sd, "%s.unit_pointnet.%d" % (pre, l), new)))
    return new


def pointnet2_local(sd, points, radii=DEFAULT_RADII, pre="encoder.local_extract", intermediates=None):
    """models/pointnet2.py:217-249.  points (B,n,3+C) -> (B,n,512)."""
    xyz, feat = P.separate_xyz_and_features(points)                           # :228
    xyz_list, feat_list = [xyz], [feat]
    for l, (M, _) in enumerate(SA_SPECS):                                     # :232
        xyz, feat = set_abstraction(sd, "%s.set_abstractions.%d" % (pre, l), xyz, feat, M,
                                    [radii[l], radii[l + 1]], intermediates)
        xyz_list.append(xyz)
        feat_list.append(feat)
    target = -2
    for l in range(5):                                                        # :238
        feat_list[target] = feature_propagator(sd, "%s.feature_propagators.%d" % (pre, l),
                                               xyz_list[target], xyz_list[target + 1],
                                               feat_list[target], feat_list[target + 1])
        target -= 1
    x = feat_list[0]
    x = F.relu(_gn(sd, pre + ".final_layers.1", _conv(sd, pre + ".final_layers.0", x)))
    x = _conv(sd, pre + ".final_layers.3", x)                                 # :204-215,247
    return x.transpose(1, 2).contiguous()


def augment_input(spatial_in, quad=True, pairs=True):
    """models/tpointnet2.py:79-90: [xyz, xyz^2, xz, xy, yz]."""
    local_in = spatial_in
    if quad:
        local_in = torch.cat([spatial_in, spatial_in * spatial_in], dim=2)
    if pairs:
        xz = spatial_in[:, :, 0:1] * spatial_in[:, :, 2:3]
        xy = spatial_in[:, :, 0:1] * spatial_in[:, :, 1:2]
        yz = spatial_in[:, :, 2:3] * spatial_in[:, :, 1:2]
        local_in = torch.cat([local_in, xz, xy, yz], dim=2)
    return local_in


def encode(sd, x, radii=DEFAULT_RADII, regress_tnocs=True, intermediates=None):
    """models/tpointnet2.py:70-115 (TPointNet2.forward).  x (B,T,N,4) -> z0 (B,1600), tnocs (B,T,N,4)."""
    B, T, N, _ = x.shape
    global_input = x.reshape(B, T * N, 4).transpose(2, 1).contiguous()        # :75
    global_feat = pointnet_global(sd, global_input)                           # :76
    spatial_in = x.reshape(B * T, N, 4)[:, :, :3]                             # :79
    local_in = augment_input(spatial_in)
    local_feat = pointnet2_local(sd, local_in, radii, intermediates=intermediates)
    local_feat = local_feat.view(B, T * N, -1).transpose(2, 1).contiguous()   # :92-93
    feat = torch.cat([local_feat, global_feat], dim=1)                        # :96
    feat = F.relu(_gn(sd, "encoder.bn1", _conv(sd, "encoder.conv1", feat)))   # :99
    feat = _gn(sd, "encoder.bn2", _conv(sd, "encoder.conv2", feat))           # :100
    tnocs = None
    if regress_tnocs:
        t = _conv(sd, "encoder.conv3", F.relu(feat))                          # :105
        tnocs = torch.sigmoid(t[:, :4, :]).transpose(2, 1).contiguous().view(B, T, N, 4)
    z0 = torch.max(feat, 2)[0]                                                # :111
    return z0, tnocs


# ----------------------------------------------------------------------------------------------
# integrators
# ----------------------------------------------------------------------------------------------
def _axpy(ys, a, ks):
    return tuple(y + a * k for y, k in zip(ys, ks))


def rk4_solve(func, y0, t0, t1, steps):
    """Classic fixed-step RK4 over [t0,t1] on a tuple state; func(t, ys) -> tuple.  4*steps evals."""
    ys = tuple(y0)
    h = (t1 - t0) / steps
    for s in range(steps):
        t = t0 + s * h
        k1 = func(t, ys)
        k2 = func(t + 0.5 * h, _axpy(ys, 0.5 * h, k1))
        k3 = func(t + 0.5 * h, _axpy(ys, 0.5 * h, k2))
        k4 = func(t + h, _axpy(ys, h, k3))
        ys = tuple(y + (h / 6.0) * (a + 2.0 * b + 2.0 * c + d) for y, a, b, c, d in zip(ys, k1, k2, k3, k4))
    return ys


# Dormand-Prince 5(4) tableau (Shampine variant used by torchdiffeq 0.0.1 dopri5.py)
_DP_ALPHA = [1 / 5, 3 / 10, 4 / 5, 8 / 9, 1.0, 1.0]
_DP_BETA = [
    [1 / 5],
    [3 / 40, 9 / 40],
    [44 / 45, -56 / 15, 32 / 9],
    [19372 / 6561, -25360 / 2187, 64448 / 6561, -212 / 729],
    [9017 / 3168, -355 / 33, 46732 / 5247, 49 / 176, -5103 / 18656],
    [35 / 384, 0, 500 / 1113, 125 / 192, -2187 / 6784, 11 / 84],
]
_DP_CSOL = [35 / 384, 0, 500 / 1113, 125 / 192, -2187 / 6784, 11 / 84, 0]
_DP_CERR = [35 / 384 - 1951 / 21600, 0, 500 / 1113 - 22642 / 50085, 125 / 192 - 451 / 720,
            -2187 / 6784 - -12231 / 42400, 11 / 84 - 649 / 6300, -1.0 / 60.0]
_DP_CMID = [6025192743 / 30085553152 / 2, 0, 51252292925 / 65400821598 / 2, -2691868925 / 45128329728 / 2,
            187940372067 / 1594534317056 / 2, -1776094331 / 19743644256 / 2, 11237099 / 235043384 / 2]


def _rms(x):
    return float(x.norm() / (x.numel() ** 0.5))


def dopri5_solve(func, y0, times, rtol, atol, counter=None):
    """Adaptive Dormand-Prince restatement of torchdiffeq 0.0.1: per-tensor
    tol = atol + rtol*max(|y0|,|y1|), ratio = mean((err/tol)^2) over the WHOLE tensor, accept iff all
    ratios <= 1, factor update with safety 0.9 / ifactor 10 / dfactor 0.2, integrates past each
    requested time and returns the 4th-order interpolant.  Decreasing `times` are handled by negating
    t and f, as upstream.  Returns a list (per time) of tuple states."""
    times = [float(t) for t in times]
    sign = 1.0
    if len(times) > 1 and times[0] > times[1]:
        sign = -1.0
        times = [-t for t in times]

    def f(t, ys):
        if counter is not None:
            counter[0] += 1
        out = func(sign * t, ys)
        return tuple(sign * o for o in out)

    ys = tuple(y0)
    t = times[0]
    f0 = f(t, ys)
    # _select_initial_step(order=4)
    scale = [atol + y.abs() * rtol for y in ys]
    d0 = max(_rms(y / s) for y, s in zip(ys, scale))
    d1 = max(_rms(k / s) for k, s in zip(f0, scale))
    h0 = 1e-6 if (d0 < 1e-5 or d1 < 1e-5) else 0.01 * max(_rms(y / s) / max(_rms(k / s), 1e-300) for y, k, s in zip(ys, f0, scale))
    y1 = _axpy(ys, h0, f0)
    f1 = f(t + h0, y1)
    d2 = max(_rms((b - a) / s) / h0 for a, b, s in zip(f0, f1, scale))
    h1 = max(1e-6, h0 * 1e-3) if (d1 <= 1e-15 and d2 <= 1e-15) else (0.01 / max(d1, d2)) ** (1.0 / 5.0)
    dt = min(100 * h0, h1)
    out = [ys]
    interp = None
    t0s, t1s = t, t
    for tn in times[1:]:
        while tn > t1s:
            # one adaptive attempt from (t, ys, f0)
            ks = [f0]
            for a, brow in zip(_DP_ALPHA, _DP_BETA):
                yi = tuple(y + dt * sum(b * k[i] for b, k in zip(brow, ks) if b != 0) for i, y in enumerate(ys))
                ks.append(f(t + a * dt, yi))
            ynew = tuple(y + dt * sum(c * k[i] for c, k in zip(_DP_CSOL, ks) if c != 0) for i, y in enumerate(ys))
            err = tuple(dt * sum(c * k[i] for c, k in zip(_DP_CERR, ks) if c != 0) for i in range(len(ys)))
            ratios = []
            for y_a, y_b, e in zip(ys, ynew, err):
                tol = atol + rtol * torch.max(y_a.abs(), y_b.abs())
                ratios.append(float(torch.mean((e / tol) ** 2)))
            accept = all(r <= 1 for r in ratios)
            r = max(ratios)
            if r == 0:
                dt_next = dt * 10.0
            else:
                dfactor = 1.0 if r < 1 else 0.2
                factor = max(1.0 / 10.0, min(math.sqrt(r) ** (1.0 / 5.0) / 0.9, 1.0 / dfactor))
                dt_next = dt / factor
            if accept:
                ymid = tuple(y + dt * sum(c * k[i] for c, k in zip(_DP_CMID, ks) if c != 0) for i, y in enumerate(ys))
                fa, fb = f0, ks[-1]
                interp = []
                for i in range(len(ys)):
                    A = 2 * dt * (fb[i] - fa[i]) - 8 * (ynew[i] + ys[i]) + 16 * ymid[i]
                    Bc = dt * (5 * fa[i] - 3 * fb[i]) + 18 * ys[i] + 14 * ynew[i] - 32 * ymid[i]
                    C = dt * (fb[i] - 4 * fa[i]) - 11 * ys[i] - 5 * ynew[i] + 16 * ymid[i]
                    D = dt * fa[i]
                    interp.append((A, Bc, C, D, ys[i]))
                t0s, t1s = t, t + dt
                t, ys, f0 = t + dt, ynew, ks[-1]
            dt = dt_next
        if interp is None or tn == t1s:
            out.append(ys)
        else:
            xx = (tn - t0s) / (t1s - t0s)
            out.append(tuple(((((A * xx) + Bc) * xx + C) * xx + D) * xx + E for (A, Bc, C, D, E) in interp))
    return out


# ----------------------------------------------------------------------------------------------
# latent ODE
# ----------------------------------------------------------------------------------------------
def dynamics(sd, z, pre="latent_ode.ode_func.dynamics_net"):
    """models/latent_ode_model.py:129-147: Linear-Tanh-Linear-Tanh-Linear-Tanh-Linear, autonomous."""
    h = z
    for l in (0, 2, 4):
        h = torch.tanh(F.linear(h, sd["%s.%d.weight" % (pre, l)], sd["%s.%d.bias" % (pre, l)]))
    return F.linear(h, sd[pre + ".6.weight"], sd[pre + ".6.bias"])


def latent_solve(sd, z_init, solve_t, method="rk4", steps_per_interval=4, counter=None):
    """models/latent_ode_model.py:45-70.  z_init (B,64), solve_t (Tu,) sorted -> (B,Tu,64)."""
    rel_t = (solve_t - solve_t[0]).tolist()                                   # :58
    fn = lambda t, ys: (dynamics(sd, ys[0]),)
    if method == "rk4":
        outs = [z_init]
        z = z_init
        for k in range(1, len(rel_t)):
            if counter is not None:
                counter[0] += 4 * steps_per_interval
            (z,) = rk4_solve(fn, (z,), rel_t[k - 1], rel_t[k], steps_per_interval)
            outs.append(z)
    else:  # reference: dopri5 rtol = atol = 1e-3 (latent_ode_model.py:38,83)
        sol = dopri5_solve(fn, (z_init,), rel_t, 1e-3, 1e-3, counter)
        outs = [s[0] for s in sol]
    return torch.stack(outs, dim=1)                                           # permute(1,0,2) :68


def aggregate_and_solve_latent(sd, z0, time_tensor, motion=64, **kw):
    """models/caspr.py:157-183."""
    B, T = time_tensor.shape
    solve_t, time_map = torch.unique(time_tensor, sorted=True, return_inverse=True)  # :166
    z_init, z_global = z0[:, :motion], z0[:, motion:]                         # :169-170
    pred_z = latent_solve(sd, z_init, solve_t, **kw)                          # :173
    batch_inds = torch.arange(B).view(-1, 1).repeat(1, T)                     # :175
    sample_feats = pred_z[batch_inds, time_map, :]                            # :177
    z_global = z_global.unsqueeze(1).expand(B, T, z_global.shape[1])
    return torch.cat([sample_feats, z_global], dim=2)                         # :181


# ----------------------------------------------------------------------------------------------
# point CNF
# ----------------------------------------------------------------------------------------------
def odenet(sd, pre, tc, y):
    """models/odefunc.py:98-105 + diffeq_layers.py:83-90 (ConcatSquashLinear) + softplus."""
    dx = y
    for l in range(4):
        lp = "%s.layers.%d" % (pre, l)
        gate = torch.sigmoid(F.linear(tc, sd[lp + "._hyper_gate.weight"], sd[lp + "._hyper_gate.bias"]))
        bias = F.linear(tc, sd[lp + "._hyper_bias.weight"])
        if dx.dim() == 3:
            gate, bias = gate.unsqueeze(1), bias.unsqueeze(1)
        dx = F.linear(dx, sd[lp + "._layer.weight"], sd[lp + "._layer.bias"]) * gate + bias
        if l < 3:
            dx = F.softplus(dx)
    return dx


GRAD_MODE = False   # True: keep the autograd graph through the divergence (training-step oracle, train_utils.py:173)


def odefunc(sd, pre, t, y, c, e=None):
    """models/odefunc.py:119-142.  Returns (dy, -divergence) ; divergence None when e is None."""
    tt = torch.ones(y.shape[0], 1, dtype=y.dtype) * t                         # :121
    tc = torch.cat([tt, c.view(y.shape[0], -1)], dim=1)                       # :133
    if e is None:
        return odenet(sd, pre + ".diffeq", tc, y), None
    if GRAD_MODE:   # training: the reference differentiates through divergence_approx (create_graph=True, odefunc.py:14)
        yy = y if y.requires_grad else y.detach().requires_grad_(True)
        dy = odenet(sd, pre + ".diffeq", tc, yy)
        e_dzdx = torch.autograd.grad(dy, yy, e, create_graph=True)[0]
        return dy, -(e_dzdx * e).sum(dim=-1, keepdim=True)
    with torch.enable_grad():
        yy = y.detach().requires_grad_(True)
        dy = odenet(sd, pre + ".diffeq", tc, yy)
        e_dzdx = torch.autograd.grad(dy, yy, e)[0]                            # odefunc.py:14
        div = (e_dzdx * e).sum(dim=-1, keepdim=True)                          # :26
    return dy.detach(), -div.detach()


def mbn_forward(sd, pre, x, logpx=None):
    """models/normalization.py:59-80 (eval mode: running stats)."""
    mean, var = sd[pre + ".running_mean"], sd[pre + ".running_var"]
    w, b = sd[pre + ".weight"], sd[pre + ".bias"]
    y = (x - mean) * torch.exp(-0.5 * torch.log(var + 1e-4))
    y = y * torch.exp(w) + b
    if logpx is None:
        return y
    logdet = (-0.5 * torch.log(var + 1e-4) + w).expand_as(x)                  # :103-108
    return y, logpx - logdet.sum(-1, keepdim=True)


def mbn_reverse(sd, pre, y, logpy=None):
    """models/normalization.py:82-101."""
    mean, var = sd[pre + ".running_mean"], sd[pre + ".running_var"]
    w, b = sd[pre + ".weight"], sd[pre + ".bias"]
    y = (y - b) * torch.exp(-w)
    x = y * torch.exp(0.5 * torch.log(var + 1e-4)) + mean
    if logpy is None:
        return x
    logdet = (-0.5 * torch.log(var + 1e-4) + w).expand_as(x)
    return x, logpy + logdet.sum(-1, keepdim=True)


def cnf_block(sd, pre, x, c, logpx, reverse, method, steps, e, counter=None):
    """models/cnf.py:70-128.  Integrates (x, logp) over [0,T] (or [T,0] when reverse)."""
    T_end = sd[pre + ".sqrt_end_time"] ** 2 if GRAD_MODE else float(sd[pre + ".sqrt_end_time"]) ** 2   # cnf.py:87-90
    t0, t1 = (T_end, 0.0) if reverse else (0.0, T_end)                        # :95-96
    with_div = logpx is not None and e is not None
    lp = logpx if logpx is not None else torch.zeros(*x.shape[:-1], 1, dtype=x.dtype)

    def fn(t, ys):
        if counter is not None:
            counter[0] += 1
        dy, ndiv = odefunc(sd, pre + ".odefunc", t, ys[0], c, e if with_div else None)
        return (dy, ndiv if ndiv is not None else torch.zeros_like(ys[1]))

    if method == "rk4":
        xs, lps = rk4_solve(fn, (x, lp), t0, t1, steps)
    else:  # reference: dopri5 atol = rtol = 1e-5 (flow.py:96-99)
        xs, lps = dopri5_solve(fn, (x, lp), [t0, t1], 1e-5, 1e-5)[-1]
    return xs, lps


def point_cnf(sd, x, c, logpx=None, reverse=False, method="rk4", steps=8, e=None, blocks=1, counter=None):
    """models/cnf.py:33-48 (SequentialFlow) over [MBN, CNF x blocks, MBN] (flow.py:68-72)."""
    n = blocks + 2
    order = range(n - 1, -1, -1) if reverse else range(n)
    for i in order:
        pre = "point_cnf.chain.%d" % i
        if i == 0 or i == n - 1:
            if reverse:
                r = mbn_reverse(sd, pre, x, logpx)
            else:
                r = mbn_forward(sd, pre, x, logpx)
            if logpx is None:
                x = r
            else:
                x, logpx = r
        else:
            x, lp = cnf_block(sd, pre, x, c, logpx, reverse, method, steps, e, counter)
            if logpx is not None:
                logpx = lp
    return x if logpx is None else (x, logpx)


def standard_normal_logprob(z):
    """models/utils.py:10-12."""
    return -0.5 * math.log(2 * math.pi) - z.pow(2) / 2


# ----------------------------------------------------------------------------------------------
# model surface
# ----------------------------------------------------------------------------------------------
def reconstruct(sd, x, y, timestamps=None, max_timestamp=5.0, method="rk4", cnf_steps=8,
                latent_steps=4, radii=DEFAULT_RADII, regress_tnocs=True, nfe=None):
    """models/caspr.py:269-308 with the base samples y (B,T,n,3) supplied by the caller (the
    reference draws them on the CPU generator, models/utils.py:25).  -> (y, logp_y, x, tnocs)."""
    B, T, N, _ = x.shape
    z0, tnocs = encode(sd, x, radii, regress_tnocs)
    if timestamps is None:
        all_times = x[:, :, 0, 3] / max_timestamp                              # :300
    else:
        all_times = timestamps.view(1, -1).repeat(B, 1)                        # :302
    lat_counter, cnf_counter = [0], [0]
    z = aggregate_and_solve_latent(sd, z0, all_times, method=method, steps_per_interval=latent_steps,
                                   counter=lat_counter)
    n = y.shape[2]
    Tz = z.shape[1]                      # decode() takes T from the latent codes (caspr.py:227): may exceed the observed steps
    yy = y.reshape(B * Tz, n, 3)
    logp_y = standard_normal_logprob(yy).view(B * Tz, n, -1).sum(2)            # :258
    xs = point_cnf(sd, yy, z.reshape(B * Tz, -1), None, True, method, cnf_steps, counter=cnf_counter)
    if nfe is not None:
        nfe[:] = [lat_counter[0], cnf_counter[0]]
    return y, logp_y.view(B, Tz, n), xs.view(B, Tz, n, 3), tnocs


def forward_nll(sd, x, sample_points, e, method="rk4", cnf_steps=8, latent_steps=4, radii=DEFAULT_RADII):
    """models/caspr.py:76-146 (eval-mode statistics).  -> (recon_loss (B,T,N), tnocs_loss (B,T,N,4))."""
    z0, tnocs = encode(sd, x, radii)
    B, T, N, _ = sample_points.shape
    tnocs_loss = (tnocs - sample_points).abs()                                 # L1Loss(reduce=False)
    all_times = sample_points[:, :, 0, 3]                                      # :106
    z = aggregate_and_solve_latent(sd, z0, all_times, method=method, steps_per_interval=latent_steps)
    pts = sample_points.reshape(B * T, N, 4)[:, :, :3].clone()
    yy, dlogp = point_cnf(sd, pts, z.reshape(B * T, -1), torch.zeros(B * T, N, 1, dtype=pts.dtype), False, method, cnf_steps, e)
    log_py = standard_normal_logprob(yy).sum(2)                                # :133-134
    log_px = log_py - dlogp.view(B * T, N)                                     # :136-138
    return (-log_px).view(B, T, -1), tnocs_loss


def training_loss(sd, x, sample_points, e, cnf_steps=8, latent_steps=4, cnf_loss_weight=0.01, tnocs_loss_weight=100.0,
                  radii=DEFAULT_RADII):
    """The scalar `run_one_epoch` back-propagates (train_utils.py:151-173), differentiable in every entry of `sd` that
    requires grad (the gradient of the discrete RK4 map, including sqrt_end_time).  -> (loss, recon_loss, tnocs_loss)."""
    global GRAD_MODE
    prev, GRAD_MODE = GRAD_MODE, True
    try:
        recon, tl = forward_nll(sd, x, sample_points, e, "rk4", cnf_steps, latent_steps, radii)
    finally:
        GRAD_MODE = prev
    return cnf_loss_weight * recon.sum(2).mean() + tnocs_loss_weight * tl[:, :, :, :4].mean(), recon, tl


def chamfer_l2(pred, gt):
    """utils/evaluations.py:40-43: mean_i min_j + mean_j min_i of squared distances, per frame."""
    d1, d2 = P.chamfer(pred, gt)
    return d1.mean(dim=1) + d2.mean(dim=1)


def approx_emd(xyz1, xyz2):
    """utils/emd.py:5-45 -> emd_cuda.approxmatch_forward + matchcost_forward.  emd_cuda (PyTorchEMD, no pin,
    README.md:33-38) is NOT under /root/reference: this restates its published algorithm (approxmatch of Fan et al.,
    "A Point Set Generation Network") -- parity unpinned, the restatement defines the contract.
    xyz1 (B,n,3), xyz2 (B,m,3) -> cost (B) = sum_{k,l} match[k,l] * |p_k - q_l|."""
    B, n, _ = xyz1.shape
    m = xyz2.shape[1]
    d2 = ((xyz1[:, :, None, :] - xyz2[:, None, :, :]) ** 2).sum(-1)              # (B,n,m)
    dist = d2.sqrt()
    multiL, multiR = (1.0, float(n // m)) if n >= m else (float(m // n), 1.0)
    remainL = torch.full((B, n), multiL, dtype=xyz1.dtype)
    remainR = torch.full((B, m), multiR, dtype=xyz1.dtype)
    cost = torch.zeros(B, dtype=xyz1.dtype)
    for j in range(7, -3, -1):
        level = 0.0 if j == -2 else -(4.0 ** j)
        K = torch.exp(level * d2)
        ratioL = remainL / (1e-9 + (K * remainR[:, None, :]).sum(2))
        sumr = (K * ratioL[:, :, None]).sum(1) * remainR
        ratioR = torch.clamp(remainR / (sumr + 1e-9), max=1.0) * remainR
        remainR = torch.clamp(remainR - sumr, min=0.0)
        inc = K * ratioL[:, :, None] * ratioR[:, None, :]                        # this level's addition to the match matrix
        cost = cost + (inc * dist).sum((1, 2))
        remainL = torch.clamp(remainL - inc.sum(2), min=0.0)
    return cost
