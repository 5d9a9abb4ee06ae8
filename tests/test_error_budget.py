"""Per-stage error budget of the HIP path on the well-conditioned (dense) input, against the f64 evaluation of the
oracle's graph (north_star: T-NOCS / sampled xyz within 1e-5 of the reference).

For every stage of encode -> advect -> sample two numbers are recorded (gpurun_out/error_budget.json, copied to
profiles/ per round):
  local       : the stage run by the HIP kernels on the f64 oracle's OWN input of that stage (rounded to f32) vs the f64
                oracle's output of that stage -- the error the stage itself adds;
  accumulated : the value the HIP pipeline carries at that point vs the f64 oracle's (what the next stage inherits),
                next to the f32 CPU oracle's own accumulated distance from f64 (any f32 implementation's floor).
Asserted: every local error stays inside the bound written next to it, and the accumulated error of the path's outputs
(z0, T-NOCS, latent trajectory, sampled xyz) against f64.

Round-2 finding (profiles/r02_error_budget.json): the first set-abstraction level was the source of nearly all of the
path's error -- its inputs are absolute coordinates squared, so the per-neighbourhood GroupNorm amplifies the f32 rounding
of a large constant (1.9e-4 locally, in ANY f32 implementation: the f32 CPU oracle carries 2.7e-4).  With the first layer
on centred inputs formed from coordinate differences (csrc/sa_mlp.hip) the level is at 1e-5 and the path's outputs are
within 1-4e-6 of the f64 evaluation; the direct HIP-vs-oracle32 difference that remains (7e-6 on xyz, 8e-6 on T-NOCS) is
the f32 oracle's own distance from f64.
"""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import model as O
from oracle import point_ops as P
from caspr_amd.utils.synthetic import dense_sequences

pytestmark = pytest.mark.gpu
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def oracle_trace(sd, x, ybase, times, cnf_steps, latent_steps):
    """The oracle's encode -> advect -> sample with every stage boundary kept (same calls as oracle.model.encode /
    reconstruct, unrolled).  dtype follows sd / x."""
    tr = {}
    B, T, N, _ = x.shape
    gin = x.reshape(B, T * N, 4).transpose(2, 1).contiguous()
    gfeat = O.pointnet_global(sd, gin)
    tr["global_gmax"], tr["global_pointfeat"] = gfeat[:, :1024, 0], gfeat[:, 1024:, :].transpose(1, 2)
    local_in = O.augment_input(x.reshape(B * T, N, 4)[:, :, :3])
    xyz, feat = P.separate_xyz_and_features(local_in)
    pre = "encoder.local_extract"
    xyz_list, feat_list = [xyz], [feat]
    for l, (M, _) in enumerate(O.SA_SPECS):
        xyz, feat = O.set_abstraction(sd, "%s.set_abstractions.%d" % (pre, l), xyz, feat, M, [O.DEFAULT_RADII[l], O.DEFAULT_RADII[l + 1]])
        xyz_list.append(xyz)
        feat_list.append(feat)
        tr["sa%d" % (l + 1)] = feat.transpose(1, 2)                        # (B', M, C) point-major
    tr["xyz_list"], tr["sa_feats"] = xyz_list, list(feat_list)
    target = -2
    fl = list(feat_list)
    for l in range(5):
        fl[target] = O.feature_propagator(sd, "%s.feature_propagators.%d" % (pre, l), xyz_list[target], xyz_list[target + 1], fl[target], fl[target + 1])
        tr["fp%d" % (l + 1)] = fl[target].transpose(1, 2)
        target -= 1
    h = F.relu(O._gn(sd, pre + ".final_layers.1", O._conv(sd, pre + ".final_layers.0", fl[0])))
    local = O._conv(sd, pre + ".final_layers.3", h).transpose(1, 2).contiguous()      # (B', N, 512)
    tr["local"] = local
    lf = local.view(B, T * N, -1).transpose(2, 1)
    feat = torch.cat([lf, gfeat], dim=1)
    y1 = O._conv(sd, "encoder.conv1", feat)
    tr["head_conv1_raw"] = y1.transpose(1, 2)
    f1 = F.relu(O._gn(sd, "encoder.bn1", y1))
    y2 = O._conv(sd, "encoder.conv2", f1)
    tr["head_conv2_raw"] = y2.transpose(1, 2)
    f2 = O._gn(sd, "encoder.bn2", y2)
    tr["z0"] = torch.max(f2, 2)[0]
    t = O._conv(sd, "encoder.conv3", F.relu(f2))
    tr["tnocs"] = torch.sigmoid(t[:, :4, :]).transpose(2, 1).contiguous().view(B, T, N, 4)
    z = O.aggregate_and_solve_latent(sd, tr["z0"], times.view(1, -1).repeat(B, 1), method="rk4", steps_per_interval=latent_steps)
    tr["latent"] = z
    c = z.reshape(B * T, -1)
    yy = ybase.reshape(B * T, ybase.shape[2], 3)
    tr["cnf_x"] = O.point_cnf(sd, yy, c, None, True, "rk4", cnf_steps).view(B, T, -1, 3)
    return tr


def err(a, b):
    return float((a.detach().cpu().double() - b.detach().cpu().double()).abs().max())


@pytest.mark.parametrize("mode", ["bf16x6", "f32"])
def test_per_stage_error_budget(seeded_sd, mode):
    from caspr_amd import ops
    from caspr_amd.models import CaSPR
    from caspr_amd.models.lazy import Lazy
    dev = torch.device("cuda:0")
    prev = ops.set_matmul_mode(mode)
    try:
        with torch.no_grad():
            budget = _budget(seeded_sd, ops, CaSPR, Lazy, dev)
    finally:
        ops.set_matmul_mode(conv=prev[0], cnf=prev[1])
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    path = os.path.join(ROOT, "gpurun_out", "error_budget.json")
    allb = json.load(open(path)) if os.path.exists(path) else {}
    allb[mode] = budget
    with open(path, "w") as f:
        json.dump(allb, f, indent=1)
    bad = ["%s: local %.2e > %.1e" % (k, v["local"], v["local_bound"]) for k, v in budget.items() if v.get("local") is not None and v["local"] > v["local_bound"]]
    # what the path carries at its outputs, against the f64 evaluation: well inside the north_star's 1e-5
    for k, bound in (("z0", 1e-5), ("tnocs", 3e-6), ("latent", 1.5e-5), ("cnf_x", 4e-6)):
        if budget[k]["accumulated"] > bound:
            bad.append("%s: accumulated %.2e > %.1e" % (k, budget[k]["accumulated"], bound))
    assert not bad, "\n".join(bad)


def _budget(seeded_sd, ops, CaSPR, Lazy, dev):
    CNF_STEPS, LAT_STEPS = 8, 4
    B, T, N, NS = 1, 3, 1024, 512
    x, sp = dense_sequences(B, T, N)
    torch.manual_seed(0)
    ybase = torch.randn(B, T, NS, 3)
    times = sp[0, :, 0, 3]
    sd64 = {k: v.double() for k, v in seeded_sd.items()}
    t64 = oracle_trace(sd64, x.double(), ybase.double(), times.double(), CNF_STEPS, LAT_STEPS)
    t32 = oracle_trace(seeded_sd, x, ybase, times, CNF_STEPS, LAT_STEPS)
    m = CaSPR(cnf_rk4_steps=CNF_STEPS, latent_rk4_steps=LAT_STEPS)
    m.load_state_dict(seeded_sd)
    m = m.to(dev).eval()
    enc, le = m.encoder, m.encoder.local_extract
    f32d = lambda t_: t_.float().to(dev).contiguous()
    out = {}

    def put(name, local, acc, bound, absmax):
        out[name] = {"local": local, "local_bound": bound, "accumulated": acc, "oracle32_accumulated": err(t32[name], t64[name]),
                     "absmax": float(absmax.abs().max())}

    # ---------------- HIP pipeline, stage by stage (accumulated) and per stage on the f64 oracle's inputs (local)
    xd = x.to(dev)
    xyz, feat = ops.prep_input(xd)
    idx = le.indices(xyz)
    C = 6
    cur_xyz, cur_feat, cur_C = xyz, feat, C
    acc_feats, acc_xyz = [feat], [xyz]
    for l, sa in enumerate(le.set_abstractions):
        nxyz, nfeat = sa.run(cur_xyz, cur_feat, cur_C, None, idx["sa"][l], feat_kind=(ops.FEAT_QUAD | ops.FEAT_PAIRS) if l == 0 else 0)
        # local: the same level on the f64 oracle's input features
        if l == 0:
            loc = nfeat
        else:
            fin = f32d(t64["sa%d" % l])
            pad = (-fin.shape[2]) % 4
            fin = F.pad(fin, (0, pad)) if pad else fin
            _, loc = sa.run(cur_xyz, fin, t64["sa%d" % l].shape[2], None, idx["sa"][l])
        put("sa%d" % (l + 1), err(loc, t64["sa%d" % (l + 1)]), err(nfeat, t64["sa%d" % (l + 1)]), 1.5e-5 if l < 2 else 8e-6, t64["sa%d" % (l + 1)])
        cur_xyz, cur_feat, cur_C = nxyz, nfeat, nfeat.shape[2]
        acc_feats.append(nfeat)
        acc_xyz.append(nxyz)
    # feature propagation
    prev_acc = Lazy(acc_feats[-1], acc_feats[-1].shape[2])
    target = -2
    o64_feats = [None] + [t64["sa%d" % k] for k in range(1, 6)]              # point-major f64 features per level (level 0 = input feat)
    prev64 = t64["sa5"]
    for l, fp in enumerate(le.feature_propagators):
        skip_acc = acc_feats[target]
        skipC = 6 if target == -6 else skip_acc.shape[2]
        prev_acc = fp.run(acc_xyz[target], acc_xyz[target + 1], skip_acc, skipC, prev_acc, idx["nn"][l])
        # local: f64 oracle's skip / previous features
        if target == -6:
            skip_loc, skipC_loc = feat, 6
        else:
            skip_loc = f32d(o64_feats[6 + target])
            skipC_loc = skip_loc.shape[2]
        pl = f32d(prev64)
        loc = fp.run(acc_xyz[target], acc_xyz[target + 1], skip_loc, skipC_loc, Lazy(pl, pl.shape[2]), idx["nn"][l])
        name = "fp%d" % (l + 1)
        put(name, err(loc.materialize(), t64[name]), err(prev_acc.materialize(), t64[name]), 8e-6, t64[name])
        prev64 = t64[name]
        target -= 1
    # final layers of the local branch
    def final(prev):
        c0, gn, c3 = le.final_layers[0], le.final_layers[1], le.final_layers[3]
        y = ops.conv1x1(le._packed_final(0), c0.bias, prev.raw, in_scale=prev.scale, in_shift=prev.shift, in_relu=prev.relu)
        s, t_ = ops.gn_stats(y, c0.out_channels, gn.weight, gn.bias)
        return ops.conv1x1(le._packed_final(3), c3.bias, y, in_scale=s, in_shift=t_, in_relu=True)
    p64 = f32d(t64["fp5"])
    put("local", err(final(Lazy(p64, p64.shape[2])), t64["local"]), err(final(prev_acc), t64["local"]), 8e-6, t64["local"])
    # global PointNet
    pf, gmax = enc.global_extract.features(xd.view(B, T * N, 4))
    put("global_gmax", err(gmax, t64["global_gmax"]), err(gmax, t64["global_gmax"]), 3e-6, t64["global_gmax"])
    put("global_pointfeat", err(pf.materialize(), t64["global_pointfeat"]), err(pf.materialize(), t64["global_pointfeat"]), 3e-6, t64["global_pointfeat"])

    # head: run TPointNet2.forward with the local branch replaced by given features (local: the f64 oracle's)
    def head_with_local(local_feat):
        orig = le.run

        def fake_run(xyz_, feat_, C_, out=None, record=None, idx=None, feat_kind=0, stop_before_last=False):
            assert not stop_before_last
            out.copy_(local_feat.view(out.shape))
            return out
        le.run = fake_run
        rec = enc.record
        enc.record = []        # the recording path keeps PointNet++'s last layer apart from the head's first (no weight fold): the local
        try:                   # features given here ARE that layer's output
            return enc(xd)
        finally:
            le.run = orig
            enc.record = rec
    z0_loc, tn_loc = head_with_local(f32d(t64["local"]))
    z0_acc, tn_acc = enc(xd)
    put("z0", err(z0_loc, t64["z0"]), err(z0_acc, t64["z0"]), 1e-5, t64["z0"])
    put("tnocs", err(tn_loc, t64["tnocs"]), err(tn_acc, t64["tnocs"]), 3e-6, t64["tnocs"])
    # latent ODE
    tt = times.view(1, -1).repeat(B, 1).to(dev)
    lat_loc = m.aggregate_and_solve_latent(f32d(t64["z0"]), tt)
    lat_acc = m.aggregate_and_solve_latent(z0_acc, tt)
    put("latent", err(lat_loc, t64["latent"]), err(lat_acc, t64["latent"]), 3e-6, t64["latent"])
    # CNF (hyper networks + solve)
    yb = ybase.to(dev)
    x_loc = m.decode(f32d(t64["latent"]), NS, y=yb)[2]
    x_acc = m.decode(lat_acc, NS, y=yb)[2]
    put("cnf_x", err(x_loc, t64["cnf_x"]), err(x_acc, t64["cnf_x"]), 3e-6, t64["cnf_x"])
    # the raw 1600-wide head convolutions (not asserted: reported to size the accumulation error of K = 1600 sums)
    (w_pt, w_g, _w_fold, _b_fold), p2, _ = enc._head_weights()
    for name in ("head_conv1_raw", "head_conv2_raw"):
        out[name] = {"local": None, "local_bound": None, "accumulated": None, "oracle32_accumulated": err(t32[name], t64[name]),
                     "absmax": float(t64[name].abs().max())}
    return out
