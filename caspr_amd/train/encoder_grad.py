"""Training path of the TPointNet++ encoder: a taped forward and an explicit backward on the HIP gradient kernels.

The reference trains through torch.autograd (`loss.backward()` at train_utils.py:173 over the graph recorded by
tpointnet2.py:70-115, pointnet.py:34-46, pointnet2.py:217-249).  Here the whole encoder is ONE autograd node
(`EncoderFunction`): its forward keeps the raw conv outputs and the GroupNorm moments (the "tape"), its backward walks
the tape in reverse calling include/caspr_hip_train.h, and hands torch the gradients of every encoder parameter, so the
caller's training loop (`loss.backward(); optimizer.step()`) is unchanged.

Differences from the inference forward (tpointnet2.py / pointnet2.py of this package): set-abstraction MLPs run on
materialised neighbourhood rows (activations must outlive the kernel for the backward pass) instead of the
register-resident fused kernel; everything else issues the same kernels plus `gn_stats_train`.
"""
import torch

from .. import ops
from .. import train_ops as T
from ..models.lazy import Lazy


class _Packs:
    """Packed forward and transposed weights, rebuilt when the parameter changes (optimizer step)."""

    def __init__(self):
        self._store = {}

    def get(self, key, w, build):
        sig = (w.data_ptr(), w._version)
        hit = self._store.get(key)
        if hit is not None and hit[0] == sig:
            return hit[1]
        val = build()
        self._store[key] = (sig, val)
        return val

    def fwd(self, conv):
        return self.get((id(conv), "f"), conv.weight, lambda: ops.PackedWeight(conv.weight.detach().reshape(conv.weight.shape[0], -1).contiguous()))

    def bwd(self, conv):
        return self.get((id(conv), "t"), conv.weight, lambda: ops.PackedWeight(conv.weight.detach().reshape(conv.weight.shape[0], -1).t().contiguous()))


class _Grads:
    """Gradient store keyed by parameter object."""

    def __init__(self):
        self.by_param = {}

    def new(self, p, shape=None):
        g = torch.empty(p.shape if shape is None else shape, device=p.device, dtype=torch.float32)
        self.by_param[id(p)] = g
        return g

    def conv(self, conv):
        """(dW as (Cout,Cin) view, dbias)"""
        gw = self.new(conv.weight)
        gb = self.new(conv.bias)
        return gw.view(gw.shape[0], -1), gb

    def gn(self, gn):
        return self.new(gn.weight), self.new(gn.bias)


# ---------------------------------------------------------------------------------------------
# conv -> GroupNorm(16) -> ReLU over whole clouds (statistics per batch entry over all its points)
# ---------------------------------------------------------------------------------------------
def _conv_gn_fwd(packs, conv, gn, cur, want_max=False, out=None, bbias=None, pw=None, bias="conv", in_relu_from=0):
    C = conv.out_channels
    res = ops.conv1x1_gn(pw if pw is not None else packs.fwd(conv), conv.bias if bias == "conv" else None, cur.raw, gn.weight, gn.bias,
                         want_max=want_max, want_moments=True, bbias=bbias, in_scale=cur.scale, in_shift=cur.shift, in_relu=cur.relu,
                         in_relu_from=in_relu_from, out=out)
    y, st = res[0], res[1:]
    rec = {"x": cur, "y": y, "mean": st[2], "rstd": st[3], "conv": conv, "gn": gn, "scale": st[0], "shift": st[1]}
    return Lazy(y, C, st[0], st[1], True), rec, (st[4] if want_max else None)


def _conv_gn_bwd(packs, grads, rec, da, relu=True, dmax=None, amax=None, need_dx=True, out=None):
    """da: gradient w.r.t. the (rectified) normalised output, dense (B,P,>=C) or None.  Returns the gradient w.r.t.
    the activated input of the conv (B,P,roundup4(Cin)), or None."""
    conv, gn = rec["conv"], rec["gn"]
    C, cin = conv.out_channels, conv.in_channels
    dg, db_ = grads.gn(gn)
    dy = T.gn_bwd(rec["y"], da, C, rec["mean"], rec["rstd"], gn.weight, gn.bias, dg, db_, relu=relu, dmax=dmax, amax=amax, out=out)
    dW, dbias = grads.conv(conv)
    x = rec["x"]
    T.conv1x1_wgrad(dy, x.raw, cin, C, dW, dbias, in_scale=x.scale, in_shift=x.shift, in_relu=x.relu)
    if not need_dx:
        return None
    return ops.conv1x1(packs.bwd(conv), None, dy)


# ---------------------------------------------------------------------------------------------
# one scale of a set-abstraction level on neighbourhood rows
# ---------------------------------------------------------------------------------------------
def _sa_scale_fwd(packs, pn, xyz, new_xyz, feat, C, idx, out, off, need_input_grad=True, feat_kind=0):
    ns = idx.shape[2]
    # First level, 16-channel scale (one channel per GroupNorm group): the conv runs on rows centred on the neighbourhood's
    # sample 0 -- the constant it drops cancels in the normalisation, forward and backward, and the result no longer
    # carries the rounding of a constant 100x larger than what the GroupNorm keeps (see csrc/sa_mlp.hip, sa_small_kernel)
    centred = feat_kind != 0 and not need_input_grad and pn.conv_layers[0].out_channels == 16
    cur = T.group_rows(xyz, new_xyz, feat, C, idx, centred=centred, feat_kind=feat_kind if centred else 0)
    recs = []
    n_layers = len(pn.conv_layers)
    for l, (conv, gn) in enumerate(zip(pn.conv_layers, pn.bn_layers)):
        y = ops.conv1x1(packs.fwd(conv), conv.bias, cur)
        last = l == n_layers - 1
        Cl = conv.out_channels
        A, mean, rstd, arg = T.gn_rows(y, ns, Cl, gn.weight, gn.bias, relu=not last, maxout=out[:, :, off:off + Cl] if last else None)
        recs.append({"x": cur, "y": y, "mean": mean, "rstd": rstd, "arg": arg, "conv": conv, "gn": gn})
        cur = A
    seg = None
    if need_input_grad and feat is not None and C > 0:
        # deterministic backward of the gather: for every input point the neighbourhood rows that read it (ascending)
        B, n = xyz.shape[0], xyz.shape[1]
        base = (torch.arange(B, device=idx.device, dtype=torch.int64) * n).view(B, 1, 1)
        seg = T.Segments(idx.long() + base, B * n)
    return {"layers": recs, "idx": idx, "ns": ns, "C": C, "off": off, "seg": seg}


def _sa_scale_bwd(packs, grads, rec, dout, dfeat):
    """dout (B,M,Ctot) gradient of the level's output; dfeat (B,n,>=C) accumulates the input-feature gradient (or None)."""
    ns, d = rec["ns"], None
    layers = rec["layers"]
    for l in range(len(layers) - 1, -1, -1):
        r = layers[l]
        conv, gn = r["conv"], r["gn"]
        Cl = conv.out_channels
        dg, db_ = grads.gn(gn)
        if l == len(layers) - 1:
            dy = T.gn_rows_bwd(r["y"], ns, Cl, gn.weight, gn.bias, False, r["mean"], r["rstd"], dg, db_,
                               dmax=dout[:, :, rec["off"]:rec["off"] + Cl], arg=r["arg"])
        else:
            dy = T.gn_rows_bwd(r["y"], ns, Cl, gn.weight, gn.bias, True, r["mean"], r["rstd"], dg, db_, da=d)
        dW, dbias = grads.conv(conv)
        T.conv1x1_wgrad(dy, r["x"], conv.in_channels, Cl, dW, dbias)
        if l > 0 or dfeat is not None:
            d = ops.conv1x1(packs.bwd(conv), None, dy)
    if dfeat is not None:
        T.segment_sum(d, rec["seg"], rec["C"], dfeat, col0=3, accumulate=True)     # fixed-order form of group_rows_bwd


# ---------------------------------------------------------------------------------------------
class EncoderTape:
    pass


def encoder_forward(enc, x):
    """Taped forward of TPointNet2 (tpointnet2.py:70-115).  x (B,T,N,4) -> z0 (B,F), tnocs (B,T,N,4) | None, tape."""
    if not x.is_cuda:
        raise ValueError("caspr_amd.TPointNet2 runs on the GPU only (HIP kernels); got a %s tensor" % x.device)
    if not hasattr(enc, "_train_packs"):
        enc._train_packs = _Packs()
    packs = enc._train_packs
    tape = EncoderTape()
    B, T_, N, _ = x.shape
    x = x.contiguous().float()
    L, S, Gf = enc.local_feat_size, enc.space_time_pt_feat, enc.global_feat_size
    P = T_ * N
    dev = x.device
    X1 = torch.empty(B, P, L + S, device=dev, dtype=torch.float32)
    xyz, feat = ops.prep_input(x, enc.augment_quad, enc.augment_pairs)
    C0 = (3 if enc.augment_quad else 0) + (3 if enc.augment_pairs else 0)
    feat_kind0 = (ops.FEAT_QUAD if enc.augment_quad else 0) | (ops.FEAT_PAIRS if enc.augment_pairs else 0)
    if C0 == 0:
        feat = None
    pn2 = enc.local_extract
    idx = pn2.indices(xyz)

    # ---- global PointNet (pointnet.py:34-46) ----
    ge = enc.global_extract
    x_pm = x.view(B, P, 4)
    a1, r1, _ = _conv_gn_fwd(packs, ge.conv1, ge.bn1, Lazy(x_pm, 4), out=X1[:, :, L:])
    a2, r2, _ = _conv_gn_fwd(packs, ge.conv2, ge.bn2, a1)
    a3, r3, gmax = _conv_gn_fwd(packs, ge.conv3, ge.bn3, a2, want_max=True)
    tape.glob = (r1, r2, r3)

    # ---- local PointNet++ per frame (pointnet2.py:217-249) ----
    xyz_list, feat_list, ch_list = [xyz], [feat], [C0]
    tape.sa = []
    for l, sa in enumerate(pn2.set_abstractions):
        d = idx["sa"][l]
        new_xyz = d["new_xyz"]
        out = torch.empty(xyz.shape[0], sa.num_points_out, sa.get_num_features_out(), device=dev, dtype=torch.float32)
        off, scales = 0, []
        for i in range(len(sa.layers)):
            scales.append(_sa_scale_fwd(packs, sa.pointnet_modules[i], xyz_list[-1], new_xyz, feat_list[-1], ch_list[-1], d["ball_idx"][i], out, off,
                                        need_input_grad=l > 0,    # level 0 reads the input coordinates' features: no gradient
                                        feat_kind=feat_kind0 if l == 0 else 0))
            off += sa.pointnet_layer_dims_list[i][-1]
        tape.sa.append(scales)
        xyz_list.append(new_xyz)
        feat_list.append(out)
        ch_list.append(out.shape[2])
    prev = Lazy(feat_list[-1], ch_list[-1])
    tape.fp = []
    target = -2
    for l, fp in enumerate(pn2.feature_propagators):
        _, nidx, w = idx["nn"][l]
        skip, Cs = feat_list[target], ch_list[target]
        xin = ops.three_interpolate(prev.raw, nidx, w, skip=skip, skip_channels=Cs, in_scale=prev.scale, in_shift=prev.shift,
                                    in_relu=prev.relu, C=prev.channels)
        cur = Lazy(xin, prev.channels + Cs)
        recs = []
        for k in range(len(fp.layer_dims)):
            cur, r, _ = _conv_gn_fwd(packs, fp.unit_pointnet[3 * k], fp.unit_pointnet[3 * k + 1], cur)
            recs.append(r)
        nB, nn_, m_ = nidx.shape[0], nidx.shape[1], prev.raw.shape[1]
        base = (torch.arange(nB, device=dev, dtype=torch.int64) * m_).view(nB, 1, 1)
        rows = (torch.arange(nB * nn_, device=dev, dtype=torch.int64).view(nB, nn_, 1)).expand(nB, nn_, 3)
        seg = T.Segments(nidx.long() + base, nB * m_, weight=w, src_rows=rows)    # deterministic backward of three_interpolate
        tape.fp.append({"layers": recs, "seg": seg, "Cprev": prev.channels, "Cs": Cs, "level": len(feat_list) + target, "m": m_})
        prev = cur
        target -= 1
    c0, gnf, c3 = pn2.final_layers[0], pn2.final_layers[1], pn2.final_layers[3]
    af, rf, _ = _conv_gn_fwd(packs, c0, gnf, prev)
    ops.conv1x1(packs.fwd(c3), c3.bias, af.raw, in_scale=af.scale, in_shift=af.shift, in_relu=True, out=X1.view(B * T_, N, L + S)[:, :, :L])
    tape.final = (rf, af, c3)
    tape.sa_shapes = [(f.shape[1], f.shape[2]) for f in feat_list[1:]]

    # ---- head (tpointnet2.py:96-113) ----
    w = enc.conv1.weight

    def build_head():
        w2 = w.detach()[:, :, 0]
        w_pt = torch.cat([w2[:, :L], w2[:, L + Gf:]], dim=1).contiguous()
        w_g = w2[:, L:L + Gf].contiguous()
        return ops.PackedWeight(w_pt), ops.PackedWeight(w_g), ops.PackedWeight(w_pt.t().contiguous()), ops.PackedWeight(w_g.t().contiguous())
    w_pt, w_g, w_pt_t, w_g_t = packs.get((id(enc.conv1), "head"), w, build_head)
    bbias = ops.conv1x1(w_g, enc.conv1.bias, gmax.view(B, 1, -1)).view(B, -1)
    ones = torch.ones(B, L, device=dev, dtype=torch.float32)
    in_scale = torch.cat([ones, a1.scale], dim=1).contiguous()
    in_shift = torch.cat([torch.zeros_like(ones), a1.shift], dim=1).contiguous()
    hin = Lazy(X1, L + S, in_scale, in_shift, True)
    h1, rh1, _ = _conv_gn_fwd(packs, enc.conv1, enc.bn1, hin, bbias=bbias, pw=w_pt, bias=None, in_relu_from=L)
    h2, rh2, z0 = _conv_gn_fwd(packs, enc.conv2, enc.bn2, h1, want_max=True)
    tnocs, t_full = None, None
    if enc.regress_tnocs:
        t_full = ops.conv1x1(packs.fwd(enc.conv3), enc.conv3.bias, h2.raw, in_scale=h2.scale, in_shift=h2.shift, in_relu=True, act=1)
        tnocs = t_full[:, :, :enc.tnocs_point_size].reshape(B, T_, N, enc.tnocs_point_size)
    tape.head = (rh1, rh2, h2, t_full, gmax, (w_pt_t, w_g_t))
    tape.dims = (B, T_, N, L, S, Gf)
    tape.x_pm = x_pm
    tape.X1 = X1
    return z0, tnocs, tape


def encoder_backward(enc, tape, dz0, dtnocs):
    """Gradients of every encoder parameter given dL/dz0 (B,F) | None and dL/dtnocs (B,T,N,4) | None.
    Returns {id(param): grad}."""
    packs, grads = enc._train_packs, _Grads()
    B, T_, N, L, S, Gf = tape.dims
    P = T_ * N
    dev = tape.X1.device
    rh1, rh2, h2, t_full, gmax, (w_pt_t, w_g_t) = tape.head
    F_ = enc.conv2.out_channels

    # ---- head ----
    da2 = None
    if dtnocs is not None and enc.regress_tnocs:
        k = enc.tnocs_point_size
        t = t_full[:, :, :k]
        dy3 = torch.zeros(B, P, (k + 3) // 4 * 4, device=dev, dtype=torch.float32)
        dy3[:, :, :k] = dtnocs.reshape(B, P, k).float() * t * (1.0 - t)            # sigmoid' (tpointnet2.py:106)
        dW3, db3 = grads.conv(enc.conv3)
        T.conv1x1_wgrad(dy3, h2.raw, F_, k, dW3, db3, in_scale=h2.scale, in_shift=h2.shift, in_relu=True)
        da2 = ops.conv1x1(packs.bwd(enc.conv3), None, dy3)
    elif enc.regress_tnocs:
        grads.new(enc.conv3.weight).zero_()
        grads.new(enc.conv3.bias).zero_()
    amax2 = None
    if dz0 is not None:
        amax2 = T.argmax_points(rh2["y"], F_, rh2["scale"], rh2["shift"])
        dz0 = dz0.contiguous().float()
    if da2 is None and dz0 is None:
        raise ValueError("encoder_backward: no incoming gradient")
    out2 = None if da2 is not None else torch.empty(B, P, F_, device=dev, dtype=torch.float32)
    da1 = _conv_gn_bwd(packs, grads, rh2, da2, relu=True, dmax=dz0, amax=amax2, out=out2)
    del da2, out2
    # conv1 of the head: weight columns [local | tiled global max | point feature]; the global block reached the
    # conv as a per-sequence bias, so its gradient comes from the per-sequence column sums of dy1
    dg1, db1_ = grads.gn(enc.bn1)
    C1 = enc.conv1.out_channels
    dy1 = T.gn_bwd(rh1["y"], da1, C1, rh1["mean"], rh1["rstd"], enc.bn1.weight, enc.bn1.bias, dg1, db1_, relu=True)
    hin = rh1["x"]
    dWpt = torch.empty(C1, L + S, device=dev, dtype=torch.float32)
    T.conv1x1_wgrad(dy1, hin.raw, L + S, C1, dWpt, None, in_scale=hin.scale, in_shift=hin.shift, in_relu=True, in_relu_from=L)
    dbb = T.colsum_batched(dy1, C1)
    dWg = torch.empty(C1, Gf, device=dev, dtype=torch.float32)
    dbias1 = grads.new(enc.conv1.bias)
    T.conv1x1_wgrad(dbb.view(B, 1, C1), gmax.view(B, 1, Gf), Gf, C1, dWg, dbias1)
    gw1 = grads.new(enc.conv1.weight).view(C1, -1)
    gw1[:, :L] = dWpt[:, :L]
    gw1[:, L:L + Gf] = dWg
    gw1[:, L + Gf:] = dWpt[:, L:]
    dgmax = ops.conv1x1(w_g_t, None, dbb.view(B, 1, C1)).view(B, -1)[:, :Gf].contiguous()
    dX1 = ops.conv1x1(w_pt_t, None, dy1)                                            # (B,P,L+S): [d local | d point feature]
    del dy1, da1

    # ---- global PointNet ----
    ge = enc.global_extract
    r1, r2, r3 = tape.glob
    amax3 = T.argmax_points(r3["y"], Gf, r3["scale"], r3["shift"])
    dy3g = torch.empty(B, P, Gf, device=dev, dtype=torch.float32)
    d2 = _conv_gn_bwd(packs, grads, r3, None, relu=False, dmax=dgmax, amax=amax3, out=dy3g)
    del dy3g
    d1 = _conv_gn_bwd(packs, grads, r2, d2, relu=True)
    d1 = d1[:, :, :S] if d1.shape[2] != S else d1
    d1 = (d1 + dX1[:, :, L:L + S]).contiguous()
    _conv_gn_bwd(packs, grads, r1, d1, relu=True, need_dx=False)

    # ---- local PointNet++ ----
    pn2 = enc.local_extract
    rf, af, c3 = tape.final
    dlocal = dX1.view(B * T_, N, L + S)[:, :, :L]
    dW, dbias = grads.conv(c3)
    T.conv1x1_wgrad(dlocal, af.raw, c3.in_channels, c3.out_channels, dW, dbias, in_scale=af.scale, in_shift=af.shift, in_relu=True)
    daf = ops.conv1x1(packs.bwd(c3), None, dlocal)
    dprev = _conv_gn_bwd(packs, grads, rf, daf, relu=True)
    del daf, dX1
    dsa = [torch.zeros(B * T_, m, c, device=dev, dtype=torch.float32) for (m, c) in tape.sa_shapes]
    for l in range(len(tape.fp) - 1, -1, -1):
        rec = tape.fp[l]
        d = dprev
        for r in reversed(rec["layers"]):
            d = _conv_gn_bwd(packs, grads, r, d, relu=True)
        Cprev, Cs, level = rec["Cprev"], rec["Cs"], rec["level"]
        if l > 0:
            dcoarse = torch.zeros(B * T_, rec["m"], Cprev, device=dev, dtype=torch.float32)
        else:
            dcoarse = dsa[-1]                                                       # FP level 0 interpolates the last SA output
        T.segment_sum(d, rec["seg"], Cprev, dcoarse, col0=0, accumulate=True)        # fixed-order form of three_interp_bwd
        if level >= 1:                                                              # skip connection from SA level `level`
            dsa[level - 1] += d[:, :, Cprev:Cprev + Cs]
        dprev = dcoarse
    for l in range(len(tape.sa) - 1, -1, -1):
        for sc in tape.sa[l]:
            _sa_scale_bwd(packs, grads, sc, dsa[l], dsa[l - 1] if l > 0 else None)
    return grads.by_param


class EncoderFunction(torch.autograd.Function):
    """z0, tnocs = EncoderFunction.apply(enc, x, *enc.parameters())"""

    @staticmethod
    def forward(ctx, enc, x, *params):
        with torch.no_grad():
            z0, tnocs, tape = encoder_forward(enc, x)
        ctx.enc, ctx.tape, ctx.params = enc, tape, params
        if tnocs is None:
            tnocs = z0.new_zeros(0)
        return z0, tnocs

    @staticmethod
    def backward(ctx, dz0, dtnocs):
        enc, tape = ctx.enc, ctx.tape
        if dtnocs is not None and dtnocs.numel() == 0:
            dtnocs = None
        with torch.no_grad():
            g = encoder_backward(enc, tape, dz0, dtnocs)
        ctx.tape = None
        out = []
        for p in ctx.params:
            gp = g.get(id(p))
            out.append(gp if (gp is not None and p.requires_grad) else None)
        return (None, None) + tuple(out)


def encode_with_grad(enc, x):
    """Differentiable encoder call: (z0, tnocs | None) carrying grad_fn."""
    params = [p for p in enc.parameters()]
    z0, tnocs = EncoderFunction.apply(enc, x, *params)
    return z0, (tnocs if enc.regress_tnocs else None)
