"""Training path of the latent ODE and the point CNF.

The reference back-propagates through torchdiffeq's adjoint (cnf.py:102-110, latent_ode_model.py:98).  This build
integrates with fixed-step RK4, so the training path differentiates the discrete RK4 map itself
(discretise-then-optimise): the gradient is exact for the map the forward pass computes, including d/d(sqrt_end_time)
through the step size and the stage times.

Status (round 1): every matrix product of the path -- the 512x512 layers of the ODE function on value AND tangent
columns (the Hutchinson divergence e^T (df/dy) e is carried as a forward-mode tangent, odefunc.py:13-31), the hyper
networks, the latent dynamics -- runs on the HIP kernels through `LinearRows` (forward: caspr_conv1x1_f32, data
gradient: the same kernel with the transposed packed weight, weight/bias gradient: caspr_conv1x1_wgrad_f32).  The
gated softplus layers (value and tangent rows, forward and backward with the per-frame gate / bias reductions) are the
fused kernels caspr_cnf_act_f32 / caspr_cnf_act_bwd_f32 (`CnfAct`).  torch.autograd records only the small tensors
around them: the (frames, C) gates, the (BT,n,3) RK4 combinations and the 3-channel output layer.  Fusing the
activation into the GEMM epilogue and recomputing instead of storing the layer products is the next step
(DESIGN.md section 7).  No CPU path: everything below requires GPU tensors.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from .. import train_ops as T

_pack_cache = {}


def _packed(w, transposed):
    """PackedWeight of a 2-D weight; cached for nn.Parameters (keyed on storage + version), fresh for temporaries."""
    def build():
        src = w.detach()
        return ops.PackedWeight((src.t() if transposed else src).contiguous())
    if not isinstance(w, nn.Parameter):
        return build()
    key = (id(w), transposed)
    sig = (w.data_ptr(), w._version)
    hit = _pack_cache.get(key)
    if hit is not None and hit[0] == sig:
        return hit[1]
    pw = build()
    _pack_cache[key] = (sig, pw)
    return pw


def _pad4(x):
    c = x.shape[-1]
    return x.contiguous() if c % 4 == 0 else F.pad(x, (0, (-c) % 4)).contiguous()


class LinearRows(torch.autograd.Function):
    """y = x W^T (+ b) over rows.  x (R, >=Cin) f32 GPU, W (Cout, Cin), b (Cout) | None  ->  (R, Cout)."""

    @staticmethod
    def forward(ctx, x, w, b):
        if not x.is_cuda:
            raise ValueError("LinearRows runs on the GPU only (HIP kernels)")
        cout, cin = w.shape
        xp = _pad4(x[:, :cin]) if x.shape[1] != (cin + 3) // 4 * 4 else x.contiguous()
        y = ops.conv1x1(_packed(w, False), None if b is None else b.detach().contiguous(), xp.view(1, xp.shape[0], xp.shape[1]))
        ctx.save_for_backward(xp, w)
        ctx.has_bias = b is not None
        ctx.x_cols = x.shape[1]
        return y.view(xp.shape[0], -1)[:, :cout]

    @staticmethod
    def backward(ctx, dy):
        xp, w = ctx.saved_tensors
        cout, cin = w.shape
        R = xp.shape[0]
        dyp = _pad4(dy)
        dyv = dyp.view(1, R, dyp.shape[1])
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = ops.conv1x1(_packed(w, True), None, dyv).view(R, -1)
            if dx.shape[1] != cin:
                dx[:, cin:] = 0.0
            if dx.shape[1] != ctx.x_cols:
                dx = dx[:, :ctx.x_cols] if dx.shape[1] > ctx.x_cols else F.pad(dx, (0, ctx.x_cols - dx.shape[1]))
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            dw = torch.empty(cout, cin, device=xp.device, dtype=torch.float32)
            db = torch.empty(cout, device=xp.device, dtype=torch.float32) if ctx.has_bias else None
            T.conv1x1_wgrad(dyv, xp.view(1, R, xp.shape[1]), cin, cout, dw, db)
        return dx, dw, db


def linear_rows(x, w, b=None):
    return LinearRows.apply(x, w, b)


class CnfAct(torch.autograd.Function):
    """Gated softplus layer on value + tangent rows (caspr_cnf_act_f32 / caspr_cnf_act_bwd_f32).
    z (2R,C) = layer product, b (C), gate / beta (frames, C), n points per frame -> h (2R,C)."""

    @staticmethod
    def forward(ctx, z, b, gate, beta, n, blk):
        from .. import lib as _lib
        from ..ops import _p, _stream
        R2, C = z.shape
        R = R2 // 2
        if not (z.is_cuda and z.stride(1) == 1 and z.stride(0) % 4 == 0):
            raise ValueError("CnfAct: z must be a GPU (2R,C) tensor with unit column stride")
        b, gate, beta = b.detach().contiguous(), gate.detach().contiguous(), beta.detach().contiguous()
        h = torch.empty(R2, C, device=z.device, dtype=torch.float32)
        _lib.check(_lib.load().caspr_cnf_act_f32(_p(z), z.stride(0), _p(b), _p(gate), _p(beta), R, n, C, blk, _p(h), C, _stream()), "caspr_cnf_act_f32")
        ctx.save_for_backward(z, b, gate, beta)
        ctx.n, ctx.blk = n, blk
        return h

    @staticmethod
    def backward(ctx, dh):
        from .. import lib as _lib
        from ..ops import _p, _stream
        z, b, gate, beta = ctx.saved_tensors
        R2, C = z.shape
        R = R2 // 2
        dh = dh.contiguous()
        dz = torch.empty(R2, C, device=z.device, dtype=torch.float32)
        dgate = torch.empty_like(gate)
        dbeta = torch.empty_like(beta)
        _lib.check(_lib.load().caspr_cnf_act_bwd_f32(_p(z), z.stride(0), _p(b), _p(gate), _p(beta), _p(dh), C, R, ctx.n, C, ctx.blk, _p(dz), C,
                                                     _p(dgate), _p(dbeta), _stream()), "caspr_cnf_act_bwd_f32")
        db = (gate * dbeta).sum(dim=0)                                    # d/db = sum_r da*g = sum_f g[f]*dbeta[f]
        return dz, db, dgate, dbeta, None, None


class CnfLayer(torch.autograd.Function):
    """A hidden layer of the ODE function in ONE launch: z = x W^T on the bf16x6 conv kernel with the gated softplus of CnfAct in
    its epilogue (caspr_conv1x1_cnf_act_bf16x6_f32; row layout blk = 32).  x (2R, Cin), W (Cout, Cin), b (Cout), gate / beta
    (frames, Cout) -> h (2R, Cout).  Same values as linear_rows + CnfAct, one pass less over the (2R, Cout) product."""

    @staticmethod
    def forward(ctx, x, w, b, gate, beta, n):
        from .. import lib as _lib
        from ..ops import _p, _stream
        cout, cin = w.shape
        R2 = x.shape[0]
        xp = x.contiguous()
        b, gate, beta = b.detach().contiguous(), gate.detach().contiguous(), beta.detach().contiguous()
        z = torch.empty(R2, cout, device=x.device, dtype=torch.float32)
        h = torch.empty(R2, cout, device=x.device, dtype=torch.float32)
        pw = _packed(w, False)
        with ops.timed("k:conv1x1_bf16x6:%d:%d:%d" % (cin, cout, R2), 2):
            _lib.check(_lib.load().caspr_conv1x1_cnf_act_bf16x6_f32(_p(pw.x3()), _p(b), _p(gate), _p(beta), _p(xp), xp.stride(0), _p(z), cout, _p(h), cout,
                                                                    R2 // (2 * n), n, cin, cout, _stream()), "caspr_conv1x1_cnf_act_bf16x6_f32")
        ctx.save_for_backward(xp, w, z, b, gate, beta)
        ctx.n = n
        return h

    @staticmethod
    def backward(ctx, dh):
        from .. import lib as _lib
        from ..ops import _p, _stream
        xp, w, z, b, gate, beta = ctx.saved_tensors
        cout, cin = w.shape
        R2 = xp.shape[0]
        dh = dh.contiguous()
        dz = torch.empty(R2, cout, device=z.device, dtype=torch.float32)
        dgate, dbeta = torch.empty_like(gate), torch.empty_like(beta)
        _lib.check(_lib.load().caspr_cnf_act_bwd_f32(_p(z), cout, _p(b), _p(gate), _p(beta), _p(dh), cout, R2 // 2, ctx.n, cout, 32, _p(dz), cout,
                                                     _p(dgate), _p(dbeta), _stream()), "caspr_cnf_act_bwd_f32")
        dzv = dz.view(1, R2, cout)
        dx = ops.conv1x1(_packed(w, True), None, dzv).view(R2, -1) if ctx.needs_input_grad[0] else None
        dw = torch.empty(cout, cin, device=xp.device, dtype=torch.float32)
        T.conv1x1_wgrad(dzv, xp.view(1, R2, cin), cin, cout, dw, None)
        return dx, dw, (gate * dbeta).sum(dim=0), dgate, dbeta, None


class CnfLayerOut(torch.autograd.Function):
    """The last hidden layer AND the 3-channel output product behind it (odefunc.py:103: no activation there): h = CnfLayer(x),
    zo = h Wo^T.  Backward: the output layer's data gradient dzo Wo is formed inside the activation's backward kernel
    (caspr_cnf_act_bwd_out_f32) instead of being written by a K = 3 conv and read back.
    x (2R, Cin), w (C, Cin), b (C), gate / beta (frames, C), wo (3, C) -> zo (2R, 3)."""

    @staticmethod
    def forward(ctx, x, w, b, gate, beta, wo, n):
        from .. import lib as _lib
        from ..ops import _p, _stream
        cout, cin = w.shape
        R2 = x.shape[0]
        xp = x.contiguous()
        b, gate, beta = b.detach().contiguous(), gate.detach().contiguous(), beta.detach().contiguous()
        z = torch.empty(R2, cout, device=x.device, dtype=torch.float32)
        h = torch.empty(R2, cout, device=x.device, dtype=torch.float32)
        with ops.timed("k:conv1x1_bf16x6:%d:%d:%d" % (cin, cout, R2), 2):
            _lib.check(_lib.load().caspr_conv1x1_cnf_act_bf16x6_f32(_p(_packed(w, False).x3()), _p(b), _p(gate), _p(beta), _p(xp), xp.stride(0), _p(z), cout,
                                                                    _p(h), cout, R2 // (2 * n), n, cin, cout, _stream()), "caspr_conv1x1_cnf_act_bf16x6_f32")
        zo = ops.conv1x1(_packed(wo, False), None, h.view(1, R2, cout)).view(R2, -1)
        ctx.save_for_backward(xp, w, z, b, gate, beta, h, wo)
        ctx.n = n
        return zo[:, :wo.shape[0]]

    @staticmethod
    def backward(ctx, dzo):
        from .. import lib as _lib
        from ..ops import _p, _stream
        xp, w, z, b, gate, beta, h, wo = ctx.saved_tensors
        cout, cin = w.shape
        R2 = xp.shape[0]
        dzop = _pad4(dzo)
        dwo = torch.empty(wo.shape[0], cout, device=xp.device, dtype=torch.float32)
        T.conv1x1_wgrad(dzop.view(1, R2, dzop.shape[1]), h.view(1, R2, cout), cout, wo.shape[0], dwo, None)
        dz = torch.empty(R2, cout, device=z.device, dtype=torch.float32)
        dgate, dbeta = torch.empty_like(gate), torch.empty_like(beta)
        woc = wo.detach().contiguous()
        _lib.check(_lib.load().caspr_cnf_act_bwd_out_f32(_p(z), cout, _p(b), _p(gate), _p(beta), _p(dzop), dzop.shape[1], _p(woc), cout, R2 // 2, ctx.n, cout,
                                                         32, _p(dz), cout, _p(dgate), _p(dbeta), _stream()), "caspr_cnf_act_bwd_out_f32")
        dzv = dz.view(1, R2, cout)
        dx = ops.conv1x1(_packed(w, True), None, dzv).view(R2, -1) if ctx.needs_input_grad[0] else None
        dw = torch.empty(cout, cin, device=xp.device, dtype=torch.float32)
        T.conv1x1_wgrad(dzv, xp.view(1, R2, cin), cin, cout, dw, None)
        return dx, dw, (gate * dbeta).sum(dim=0), dgate, dbeta, dwo, None


class SplitLayers(torch.autograd.Function):
    """(gate, bias) of all layers of one evaluation, stored LAYER-MAJOR in two 1-D tensors ([layer][frame][channel]) -> the per-layer
    (frames, C_l) tensors as contiguous VIEWS.  The gradient comes back as ONE concatenation per tensor.  (Column slices of a
    (frames, sum C) tensor cost a copy per layer and tensor on the way in and a zero-fill + copy each on the way back: 22
    launches per evaluation.)"""

    @staticmethod
    def forward(ctx, gate_lm, bias_lm, BT, widths):
        outs, off = [], 0
        for w in widths:
            outs.append(gate_lm[off:off + BT * w].view(BT, w))
            off += BT * w
        off = 0
        for w in widths:
            outs.append(bias_lm[off:off + BT * w].view(BT, w))
            off += BT * w
        ctx.BT, ctx.widths = BT, widths
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        L = len(ctx.widths)
        def cat(gs):
            gs = [g.reshape(-1) if g is not None else torch.zeros(ctx.BT * w, device=ref.device, dtype=ref.dtype) for g, w in zip(gs, ctx.widths)]
            return torch.cat(gs)
        ref = next(g for g in grads if g is not None)
        return cat(grads[:L]), cat(grads[L:]), None, None


class CnfOut(torch.autograd.Function):
    """Epilogue of the 3-channel output layer (caspr_cnf_out_f32 / _bwd_f32): zo (2R, 3) [a view of the conv's 4-wide rows], b (3),
    gate / beta (frames, 3) [column slices of the all-layer tensors: passed with their row stride], e (R, 3) ->
    a (BT, n, 3) = dy/dt and nd (BT, n, 1) = - e^T (df/dy) e.  One launch each way instead of ~25 element-wise ones."""

    @staticmethod
    def forward(ctx, zo, b, gate, beta, e_rows, n, blk):
        from .. import lib as _lib
        from ..ops import _p, _stream
        R = e_rows.shape[0]
        if zo.stride(1) != 1 or gate.stride(1) != 1 or beta.stride(1) != 1 or gate.stride(0) != beta.stride(0):
            raise ValueError("CnfOut: unit column strides and a common row stride of gate / beta are required")
        b = b.detach().contiguous()
        a = torch.empty(R, 3, device=zo.device, dtype=torch.float32)
        nd = torch.empty(R, 1, device=zo.device, dtype=torch.float32)
        _lib.check(_lib.load().caspr_cnf_out_f32(_p(zo), zo.stride(0), _p(b), _p(gate), _p(beta), gate.stride(0), _p(e_rows), R, n, blk, _p(a), _p(nd),
                                                 _stream()), "caspr_cnf_out_f32")
        ctx.save_for_backward(zo, b, gate, e_rows)
        ctx.n, ctx.blk = n, blk
        return a.view(R // n, n, 3), nd.view(R // n, n, 1)

    @staticmethod
    def backward(ctx, da, dnd):
        from .. import lib as _lib
        from ..ops import _p, _stream
        zo, b, gate, e_rows = ctx.saved_tensors
        R, n = e_rows.shape[0], ctx.n
        frames = R // n
        da, dnd = da.contiguous(), dnd.contiguous()
        dzo = torch.empty(2 * R, 4, device=zo.device, dtype=torch.float32)
        dgate = torch.empty(frames, 3, device=zo.device, dtype=torch.float32)
        dbeta = torch.empty(frames, 3, device=zo.device, dtype=torch.float32)
        _lib.check(_lib.load().caspr_cnf_out_bwd_f32(_p(da), _p(dnd), _p(zo), zo.stride(0), _p(b), _p(gate), gate.stride(0), _p(e_rows), R, n, ctx.blk,
                                                     _p(dzo), _p(dgate), _p(dbeta), _stream()), "caspr_cnf_out_bwd_f32")
        return dzo[:, :3], (gate * dbeta).sum(dim=0), dgate, dbeta, None, None, None


class CnfHidden(torch.autograd.Function):
    """Both hidden layers of the ODE function and the 3-channel output product as ONE node: forward = two fused conv + activation
    launches + the output conv; backward = the output layer's data gradient inside the last layer's activation backward (as
    CnfLayerOut), and the FIRST hidden layer's activation backward in the epilogue of the data-gradient conv that produces its dH
    (caspr_conv1x1_cnf_act_bwd_bf16x6_f32): that dH (2R x C) is never written, one 1 GB pass per evaluation less.
    x (2R, C0), (w1, b1, gate1, beta1), (w2, b2, gate2, beta2), wo (3, C2) -> zo (2R, 3).  Row layout blk = 32."""

    @staticmethod
    def forward(ctx, x, w1, b1, g1, be1, w2, b2, g2, be2, wo, n):
        from .. import lib as _lib
        from ..ops import _p, _stream
        L = _lib.load()
        R2 = x.shape[0]
        xp = x.contiguous()
        b1, g1, be1, b2, g2, be2 = [t.detach().contiguous() for t in (b1, g1, be1, b2, g2, be2)]
        zs, hs, cur = [], [], xp
        for (w, b, g, be) in ((w1, b1, g1, be1), (w2, b2, g2, be2)):
            cout, cin = w.shape
            z = torch.empty(R2, cout, device=x.device, dtype=torch.float32)
            h = torch.empty(R2, cout, device=x.device, dtype=torch.float32)
            with ops.timed("k:conv1x1_bf16x6:%d:%d:%d" % (cin, cout, R2), 2):
                _lib.check(L.caspr_conv1x1_cnf_act_bf16x6_f32(_p(_packed(w, False).x3()), _p(b), _p(g), _p(be), _p(cur), cur.stride(0), _p(z), cout, _p(h), cout,
                                                              R2 // (2 * n), n, cin, cout, _stream()), "caspr_conv1x1_cnf_act_bf16x6_f32")
            zs.append(z)
            hs.append(h)
            cur = h
        zo = ops.conv1x1(_packed(wo, False), None, cur.view(1, R2, cur.shape[1])).view(R2, -1)
        ctx.save_for_backward(xp, w1, b1, g1, be1, w2, b2, g2, be2, wo, zs[0], hs[0], zs[1], hs[1])
        ctx.n = n
        return zo[:, :wo.shape[0]]

    @staticmethod
    def backward(ctx, dzo):
        from .. import lib as _lib
        from ..ops import _p, _stream, _workspace
        L = _lib.load()
        xp, w1, b1, g1, be1, w2, b2, g2, be2, wo, z1, h1, z2, h2 = ctx.saved_tensors
        n, R2 = ctx.n, xp.shape[0]
        c1, c0 = w1.shape
        c2 = w2.shape[0]
        dev = xp.device
        dzop = _pad4(dzo)
        dwo = torch.empty(wo.shape[0], c2, device=dev, dtype=torch.float32)
        T.conv1x1_wgrad(dzop.view(1, R2, dzop.shape[1]), h2.view(1, R2, c2), c2, wo.shape[0], dwo, None)
        # last hidden layer: dH = dzo Wo formed inside the activation backward
        dz2 = torch.empty(R2, c2, device=dev, dtype=torch.float32)
        dg2, db2 = torch.empty_like(g2), torch.empty_like(be2)
        woc = wo.detach().contiguous()
        _lib.check(L.caspr_cnf_act_bwd_out_f32(_p(z2), c2, _p(b2), _p(g2), _p(be2), _p(dzop), dzop.shape[1], _p(woc), c2, R2 // 2, n, c2, 32, _p(dz2), c2,
                                               _p(dg2), _p(db2), _stream()), "caspr_cnf_act_bwd_out_f32")
        dw2 = torch.empty(c2, c1, device=dev, dtype=torch.float32)
        T.conv1x1_wgrad(dz2.view(1, R2, c2), h1.view(1, R2, c1), c1, c2, dw2, None)
        # first hidden layer: its activation backward in the epilogue of dz2 W2 (the product that is its dH)
        dz1 = torch.empty(R2, c1, device=dev, dtype=torch.float32)
        dg1, db1 = torch.empty_like(g1), torch.empty_like(be1)
        frames = R2 // (2 * n)
        nbytes = L.caspr_conv1x1_cnf_act_bwd_ws_bytes(frames, n, c1)
        ws = _workspace(nbytes, dev)
        with ops.timed("k:conv1x1_bf16x6:%d:%d:%d" % (c2, c1, R2), 2):
            _lib.check(L.caspr_conv1x1_cnf_act_bwd_bf16x6_f32(_p(_packed(w2, True).x3()), _p(dz2), c2, _p(z1), c1, _p(b1), _p(g1), _p(be1), _p(dz1), c1,
                                                              _p(dg1), _p(db1), _p(ws), ws.numel(), frames, n, c2, c1, _stream()),
                       "caspr_conv1x1_cnf_act_bwd_bf16x6_f32")
        dw1 = torch.empty(c1, c0, device=dev, dtype=torch.float32)
        T.conv1x1_wgrad(dz1.view(1, R2, c1), xp.view(1, R2, c0), c0, c1, dw1, None)
        dx = ops.conv1x1(_packed(w1, True), None, dz1.view(1, R2, c1)).view(R2, -1) if ctx.needs_input_grad[0] else None
        return (dx, dw1, (g1 * db1).sum(dim=0), dg1, db1, dw2, (g2 * db2).sum(dim=0), dg2, db2, dwo, None)


def _fused_layer_ok(l, n):
    cout, cin = l._layer.weight.shape
    return ops.CONV_BF16X6 and cin % 32 == 0 and cin >= 64 and cout % 4 == 0 and cout >= 128 and n % 64 == 0


class CnfIn(torch.autograd.Function):
    """First ODE-function layer (3 -> C) fused with gate + softplus on value / tangent rows
    (caspr_cnf_in_f32 / caspr_cnf_in_bwd_f32).  y, e (R,3); w0 (C,3); b0 (C); gate / beta (frames, C) -> h (2R, C)."""

    @staticmethod
    def forward(ctx, y, e, w0, b0, gate, beta, n, blk):
        from .. import lib as _lib
        from ..ops import _p, _stream
        if not y.is_cuda:
            raise ValueError("CnfIn runs on the GPU only (HIP kernels)")
        R, C = y.shape[0], w0.shape[0]
        y, e = y.detach().contiguous(), e.detach().contiguous()
        w0, b0, gate, beta = w0.detach().contiguous(), b0.detach().contiguous(), gate.detach().contiguous(), beta.detach().contiguous()
        h = torch.empty(2 * R, C, device=y.device, dtype=torch.float32)
        _lib.check(_lib.load().caspr_cnf_in_f32(_p(y), _p(e), _p(w0), _p(b0), _p(gate), _p(beta), R, n, C, blk, _p(h), _stream()), "caspr_cnf_in_f32")
        ctx.save_for_backward(y, e, w0, b0, gate, beta)
        ctx.n, ctx.blk = n, blk
        return h

    @staticmethod
    def backward(ctx, dh):
        from .. import lib as _lib
        from ..ops import _p, _stream
        y, e, w0, b0, gate, beta = ctx.saved_tensors
        R, C = y.shape[0], w0.shape[0]
        L = _lib.load()
        ch, ns = L.caspr_cnf_in_bwd_chunk(C), L.caspr_cnf_in_bwd_splits(C, ctx.n)
        frames, chunks = R // ctx.n, (C + ch - 1) // ch
        dh = dh.contiguous()
        dgate = torch.empty(frames, ns, C, device=y.device, dtype=torch.float32)
        dbeta = torch.empty(frames, ns, C, device=y.device, dtype=torch.float32)
        dw_part = torch.empty(frames * ns, C, 3, device=y.device, dtype=torch.float32)
        dy_part = torch.empty(chunks, R, 3, device=y.device, dtype=torch.float32)
        _lib.check(L.caspr_cnf_in_bwd_f32(_p(y), _p(e), _p(w0), _p(b0), _p(gate), _p(beta), _p(dh), R, ctx.n, C, ctx.blk, _p(dgate), _p(dbeta),
                                          _p(dw_part), _p(dy_part), _stream()), "caspr_cnf_in_bwd_f32")
        if ns > 1:
            dgate, dbeta = dgate.sum(dim=1), dbeta.sum(dim=1)
        else:
            dgate, dbeta = dgate[:, 0], dbeta[:, 0]
        return dy_part.sum(dim=0), None, dw_part.sum(dim=0), (gate * dbeta).sum(dim=0), dgate, dbeta, None, None


# ---------------------------------------------------------------------------------------------
# latent ODE (latent_ode_model.py:45-70,139-147): z' = MLP_tanh(z), classic RK4, `steps` per requested interval
# ---------------------------------------------------------------------------------------------
class LatentSolve(torch.autograd.Function):
    """The whole latent RK4 solve as ONE autograd node: a taped forward and a hand-written reverse sweep over the same kernels
    (conv1x1 forward / transposed), with the WEIGHT gradients of all evaluations taken at the end as one product per layer over
    the concatenated rows (the four layers are shared by every evaluation: dW = sum_e d_e^T x_e = [d_1; d_2; ...]^T [x_1; x_2; ...]).
    Node per layer call (`linear_rows`) this was 288 eight-row weight-gradient launches with their slab reductions and 288
    gradient accumulations per step (cfg-3: 72 evaluations x 4 layers); same arithmetic otherwise.
    z0 (B,D), tt (Tu,) device times, steps, then w0, b0, ..., w3, b3 -> (B,Tu,D)."""

    TEAM = True        # the one-launch forms (caspr_latent_rk4_team_tape_f32 / _adjoint_f32) where the shape allows

    @staticmethod
    def _team_ok(z0, ws, bs):
        return (LatentSolve.TEAM and len(ws) == 4 and all(b is not None for b in bs) and z0.shape[0] <= 64 and ws[0].shape[0] == 512
                and ws[1].shape == (512, 512) and ws[2].shape == (512, 512) and ws[3].shape[1] == 512 and ws[0].shape[1] == ws[3].shape[0] <= 64
                and ws[0].shape[1] == z0.shape[1])

    @staticmethod
    def _forward_team(ctx, z0, tt, steps, ws, bs):
        from .. import lib as _lib
        from ..ops import _p, _stream
        L = _lib.load()
        B, D = z0.shape
        Tu = tt.shape[0]
        E, xw = 4 * steps * (Tu - 1), (D + 3) // 4 * 4
        dev = z0.device
        pk = [_packed(w, False) for w in ws]
        bd = [b.detach().contiguous() for b in bs]
        zc = z0.detach().contiguous()
        tape = [torch.zeros(E * B, xw, device=dev, dtype=torch.float32)] + [torch.zeros(E * B, 512, device=dev, dtype=torch.float32) for _ in range(3)]
        out = torch.empty(B, Tu, D, device=dev, dtype=torch.float32)
        key = (dev.index, torch.cuda.current_stream().cuda_stream)
        capturing = torch.cuda.is_current_stream_capturing()       # as ops.latent_rk4: no event queries / host copies under hipGraph capture
        if not capturing:
            ops._team_raise_if_failed(key)
        wsb = ops._team_workspace(L.caspr_latent_team_ws_bytes(B), dev)
        ptrs = [p_ for w, b in zip(pk, bd) for p_ in (_p(w.data), _p(b))]
        with ops.timed("latent_rk4_tape"):
            _lib.check(L.caspr_latent_rk4_team_tape_f32(_p(zc), zc.stride(0), _p(tt), B, Tu, D, 512, int(steps), *ptrs, _p(out), _p(tape[0]), xw, _p(tape[1]),
                                                        _p(tape[2]), _p(tape[3]), _p(wsb), wsb.numel(), _stream()), "caspr_latent_rk4_team_tape_f32")
        if not capturing:
            ops._team_track(key, wsb, B)
        ctx.team, ctx.tape, ctx.tt, ctx.steps, ctx.ws = True, tape, tt, steps, ws
        return out

    @staticmethod
    def _backward_team(ctx, gout):
        from .. import lib as _lib
        from ..ops import _p, _stream
        L = _lib.load()
        tape, tt, steps, ws = ctx.tape, ctx.tt, ctx.steps, ctx.ws
        B, Tu, D = gout.shape
        xw = tape[0].shape[1]
        dev = gout.device
        pkt = [_packed(w, True) for w in ws]
        g = gout.contiguous()
        deltas = [torch.zeros_like(tape[1]) for _ in range(3)] + [torch.zeros_like(tape[0])]
        gz = torch.empty(B, D, device=dev, dtype=torch.float32)
        key = (dev.index, torch.cuda.current_stream().cuda_stream)
        capturing = torch.cuda.is_current_stream_capturing()       # as ops.latent_rk4: no event queries / host copies under hipGraph capture
        if not capturing:
            ops._team_raise_if_failed(key)
        wsb = ops._team_workspace(L.caspr_latent_team_ws_bytes(B), dev)
        with ops.timed("latent_rk4_adjoint"):
            _lib.check(L.caspr_latent_rk4_team_adjoint_f32(_p(g), _p(tt), B, Tu, D, 512, int(steps), _p(pkt[3].data), _p(pkt[2].data), _p(pkt[1].data),
                                                           _p(pkt[0].data), _p(tape[1]), _p(tape[2]), _p(tape[3]), _p(deltas[0]), _p(deltas[1]),
                                                           _p(deltas[2]), _p(deltas[3]), xw, _p(gz), _p(wsb), wsb.numel(), _stream()),
                       "caspr_latent_rk4_team_adjoint_f32")
        if not capturing:
            ops._team_track(key, wsb, B)
        grads = []
        for i in range(4):
            cout, cin = ws[i].shape
            dw = torch.empty(cout, cin, device=dev, dtype=torch.float32)
            db = torch.empty(cout, device=dev, dtype=torch.float32)
            T.conv1x1_wgrad(deltas[i].view(1, deltas[i].shape[0], -1), tape[i].view(1, tape[i].shape[0], -1), cin, cout, dw, db)
            grads += [dw, db]
        ctx.tape = None
        return (gz, None, None) + tuple(grads)

    @staticmethod
    def forward(ctx, z0, tt, steps, *wb):
        ws, bs = wb[0::2], wb[1::2]
        ctx.single = tt.shape[0] == 1
        if ctx.single:          # one time stamp: nothing to integrate (odeint returns the initial state, latent_ode_model.py:58-66)
            ctx.ws, ctx.has_b = ws, [b is not None for b in bs]
            return z0.detach().unsqueeze(1).clone()
        if LatentSolve._team_ok(z0, ws, bs):
            return LatentSolve._forward_team(ctx, z0, tt, steps, ws, bs)
        ctx.team = False
        pk = [_packed(w, False) for w in ws]
        bd = [b.detach().contiguous() for b in bs]
        B = z0.shape[0]
        tape = []

        def f(z):
            a = [z.contiguous()]
            for i in range(4):
                y = ops.conv1x1(pk[i], bd[i], a[-1].view(1, B, -1)).view(B, -1)[:, :ws[i].shape[0]]
                a.append(torch.tanh(y) if i < 3 else y)
            tape.append(a[:4])
            return a[4]
        outs, z, hs = [z0], z0.detach(), []
        for k in range(1, tt.shape[0]):
            h = (tt[k] - tt[k - 1]) / steps
            hh = (h, 0.5 * h, h / 3.0, h / 6.0)         # 0-dim device tensors, made once per interval: one fused launch per RK4 combination
            for _ in range(steps):
                k1 = f(z)
                k2 = f(torch.addcmul(z, k1, hh[1]))
                k3 = f(torch.addcmul(z, k2, hh[1]))
                k4 = f(torch.addcmul(z, k3, hh[0]))
                z = torch.addcmul(z, torch.add(k1 + k4, k2 + k3, alpha=2.0), hh[3])
                hs.append(hh)
            outs.append(z)
        ctx.tape, ctx.hs, ctx.steps, ctx.ws = tape, hs, steps, ws
        ctx.has_b = [b is not None for b in bs]
        return torch.stack(outs, dim=1)

    @staticmethod
    def backward(ctx, gout):
        if ctx.single:
            zeros = [g for w, hb in zip(ctx.ws, ctx.has_b) for g in (torch.zeros_like(w), torch.zeros(w.shape[0], device=w.device, dtype=w.dtype) if hb else None)]
            return (gout[:, 0].contiguous(), None, None) + tuple(zeros)
        if ctx.team:
            return LatentSolve._backward_team(ctx, gout)
        tape, hs, steps, ws = ctx.tape, ctx.hs, ctx.steps, ctx.ws
        pkt = [_packed(w, True) for w in ws]
        B = gout.shape[0]
        deltas, inputs = [[] for _ in range(4)], [[] for _ in range(4)]

        def fb(a, g):                              # a = (x, h1, h2, h3) of one evaluation, g = dL/d f(x) -> dL/dx
            for i in (3, 2, 1, 0):
                d = g.contiguous() if i == 3 else torch.ops.aten.tanh_backward(g, a[i + 1])    # tanh' from the layer's stored output: one launch
                deltas[i].append(d)
                inputs[i].append(a[i])
                g = ops.conv1x1(pkt[i], None, d.view(1, B, -1)).view(B, -1)[:, :ws[i].shape[1]]
            return g
        Tu = gout.shape[1]
        gz = gout[:, Tu - 1].clone()
        e = len(tape)
        for k in range(Tu - 1, 0, -1):
            for st in range(steps):
                h, h2, h3, h6 = hs[(k - 1) * steps + (steps - 1 - st)]
                a1, a2, a3, a4 = tape[e - 4], tape[e - 3], tape[e - 2], tape[e - 1]
                e -= 4
                g6, g3z = h6 * gz, h3 * gz
                g4 = fb(a4, g6)
                g3 = fb(a3, torch.addcmul(g3z, g4, h))
                g2 = fb(a2, torch.addcmul(g3z, g3, h2))
                g1 = fb(a1, torch.addcmul(g6, g2, h2))
                gz = gz + (g4 + g3) + (g2 + g1)
            gz = gz + gout[:, k - 1]
        grads = []
        for i in range(4):
            d, x = torch.cat(deltas[i], dim=0), torch.cat(inputs[i], dim=0)
            cout, cin = ws[i].shape
            dw = torch.empty(cout, cin, device=d.device, dtype=torch.float32)
            db = torch.empty(cout, device=d.device, dtype=torch.float32) if ctx.has_b[i] else None
            T.conv1x1_wgrad(_pad4(d).view(1, d.shape[0], -1), _pad4(x).view(1, x.shape[0], -1), cin, cout, dw, db)
            grads += [dw, db]
        ctx.tape = None
        return (gz, None, None) + tuple(grads)


def latent_solve_layers(lat, z0, times):
    """The same solve with one autograd node per layer call (`linear_rows`) and torch.autograd's own reverse sweep: what
    LatentSolve is checked against (tests/test_hip_train.py)."""
    lin = [lat.ode_func.dynamics_net[i] for i in (0, 2, 4, 6)]

    def f(z):
        h = z
        for i, l in enumerate(lin):
            h = linear_rows(h, l.weight, l.bias)
            if i < 3:
                h = torch.tanh(h)
        return h
    outs, z, tt = [z0], z0, times.detach().float()
    for k in range(1, tt.shape[0]):
        h = (tt[k] - tt[k - 1]) / lat.rk4_steps
        for _ in range(lat.rk4_steps):
            k1 = f(z)
            k2 = f(z + 0.5 * h * k1)
            k3 = f(z + 0.5 * h * k2)
            k4 = f(z + h * k3)
            z = z + (h / 6.0) * (k1 + 2.0 * k2 + 2.0 * k3 + k4)
        outs.append(z)
    return torch.stack(outs, dim=1)


from ..config import config as _cfg     # (A/B timing and debugging switches: caspr_amd/config.py, environment only under CASPR_DEBUG=1)
OUT_NODE = _cfg.train_cnf_out_node          # False: the output layer's epilogue in torch element-wise ops
HIDDEN_NODE = _cfg.train_cnf_hidden_node    # False: CnfLayer + CnfLayerOut
LATENT_NODE = _cfg.train_latent_node        # False: the per-layer form
CHECKPOINT_STEPS = _cfg.train_cnf_checkpoint   # True: the CNF's tape per RK4 step, recomputed in the backward pass (cnf_block_train)


def latent_solve_train(lat, z0, times):
    """z0 (B,D) differentiable, times (Tu,) ascending -> (B,Tu,D); first output is z0 itself."""
    tt = times.detach().float()
    if LATENT_NODE:
        wb = []
        for i in (0, 2, 4, 6):
            l = lat.ode_func.dynamics_net[i]
            wb += [l.weight, l.bias]
        out = LatentSolve.apply(z0, tt, lat.rk4_steps, *wb)
    else:
        out = latent_solve_layers(lat, z0, times)
    lat.ode_func._num_evals.fill_(4 * lat.rk4_steps * max(tt.shape[0] - 1, 0))
    return out


# ---------------------------------------------------------------------------------------------
# point CNF block (cnf.py:70-128, odefunc.py:119-142, diffeq_layers.py:83-90)
# ---------------------------------------------------------------------------------------------
def cnf_block_train(block, x, context, logpx, e):
    """x (BT,n,3), context (BT,zdim), logpx (BT,n,1), e (BT,n,3) fixed Hutchinson noise.  Forward direction
    t: 0 -> sqrt_end_time^2 with `block.rk4_steps` RK4 steps.  -> (x_T, logp_T), differentiable in every parameter."""
    layers = block.odefunc.diffeq.layers
    BT, n, _ = x.shape
    c = context.contiguous()
    # hyper networks: the context columns once per step (constant over the solve), the time column per evaluation.
    # All four layers' gate / bias rows live in ONE (BT, sum C) tensor so that the per-evaluation time update is a
    # handful of element-wise launches instead of five per layer.
    G, Bb, tg, tb, widths = [], [], [], [], []
    for l in layers:
        wg, wb = l._hyper_gate.weight, l._hyper_bias.weight
        G.append(linear_rows(c, wg[:, 1:].contiguous(), l._hyper_gate.bias))
        Bb.append(linear_rows(c, wb[:, 1:].contiguous(), None))
        tg.append(wg[:, 0])
        tb.append(wb[:, 0])
        widths.append(wg.shape[0])
    # layer-major 1-D tensors ([layer][frame][channel]): one element-wise update per evaluation for all layers, per-layer contiguous views
    G_lm, Bb_lm = torch.cat([g.reshape(-1) for g in G]), torch.cat([b_.reshape(-1) for b_ in Bb])
    tg_lm = torch.cat([v.unsqueeze(0).expand(BT, -1).reshape(-1) for v in tg])
    tb_lm = torch.cat([v.unsqueeze(0).expand(BT, -1).reshape(-1) for v in tb])
    widths = tuple(widths)
    e_rows = e.reshape(BT * n, 3)
    # row layout of the (2R, C) tensors of this solve (include/caspr_hip_train.h): blocks of 32 value rows + the tangent rows of the
    # same points when the hidden layers run with the activation in the conv's epilogue (CnfLayer), [values | tangents] otherwise
    fused = all(_fused_layer_ok(l, n) for l in layers[1:3])
    blk = 32 if fused else BT * n

    def func(t, y, _lp):
        R = BT * n
        h = None
        parts = SplitLayers.apply(torch.sigmoid(G_lm + t * tg_lm), Bb_lm + t * tb_lm, BT, widths)   # context part + time column
        for i, l in enumerate(layers):
            gate, bias = parts[i], parts[len(layers) + i]
            if i == 0:                                                    # 3 -> C: fused product + gate + softplus, value | tangent rows
                h = CnfIn.apply(y.reshape(R, 3), e_rows, l._layer.weight, l._layer.bias, gate, bias, n, blk)
                continue
            if i == 1 and fused and not HIDDEN_NODE:
                h = CnfLayer.apply(h, l._layer.weight, l._layer.bias, gate, bias, n)   # product + gate + softplus in one launch
                continue
            if i == 2 and fused and not HIDDEN_NODE:                      # ... and the output product behind the last hidden layer
                z = CnfLayerOut.apply(h, l._layer.weight, l._layer.bias, gate, bias, layers[3]._layer.weight, n)
                continue
            if i == 1 and fused:                                          # both hidden layers + the output product: one node
                l2 = layers[2]
                z = CnfHidden.apply(h, l._layer.weight, l._layer.bias, gate, bias, l2._layer.weight, l2._layer.bias,
                                    parts[2], parts[len(layers) + 2], layers[3]._layer.weight, n)
                continue
            if i == 2 and fused:
                continue
            if not (i == 3 and fused):
                z = linear_rows(h, l._layer.weight, None)
            if i < 3:
                h = CnfAct.apply(z, l._layer.bias, gate, bias, n, blk)    # fused gate + softplus on value / tangent rows
            else:                                                         # 512 -> 3 output layer: (BT,n,3) tensors
                if OUT_NODE:
                    a, nd = CnfOut.apply(z, l._layer.bias, gate, bias, e_rows, n, blk)    # (zv + b) g + beta,  - sum_j zt_j g_j e_j
                else:                                                     # the same in torch element-wise ops (A/B timing, debugging)
                    cout = l._layer.weight.shape[0]
                    zz = z.view(R // blk, 2, blk, cout)
                    a = (zz[:, 0].reshape(BT, n, cout) + l._layer.bias) * gate.unsqueeze(1) + bias.unsqueeze(1)
                    nd = -((zz[:, 1].reshape(BT, n, cout) * gate.unsqueeze(1)) * e).sum(dim=-1, keepdim=True)
        return a, nd

    t_end = block.sqrt_end_time * block.sqrt_end_time if block.train_T else torch.tensor(float(block.T), device=x.device)
    steps = block.rk4_steps
    hstep = t_end / steps
    def rk4_step(t, y, lp):
        k1 = func(t, y, lp)
        k2 = func(t + 0.5 * hstep, y + 0.5 * hstep * k1[0], lp)
        k3 = func(t + 0.5 * hstep, y + 0.5 * hstep * k2[0], lp)
        k4 = func(t + hstep, y + hstep * k3[0], lp)
        return (y + (hstep / 6.0) * (k1[0] + 2.0 * k2[0] + 2.0 * k3[0] + k4[0]),
                lp + (hstep / 6.0) * (k1[1] + 2.0 * k2[1] + 2.0 * k3[1] + k4[1]))

    # CHECKPOINT_STEPS (config.train_cnf_checkpoint; off by default): keep only the state (y, logp) at the RK4 step boundaries and
    # recompute a step's four evaluations when the backward pass reaches it -- what the reference's adjoint does too (cnf.py:100-110:
    # O(1) memory, the trajectory re-integrated backwards).  The tape of ALL 4 S evaluations is 63 GB at the cfg-3 shard; of one
    # step, 1 / S of that.  Costs one more forward of the block (same kernels, same bits: the step is a pure function of its inputs).
    ckpt = CHECKPOINT_STEPS and torch.is_grad_enabled()
    if ckpt:
        from torch.utils.checkpoint import checkpoint
    y, lp = x, logpx
    for s in range(steps):
        t = hstep * s
        y, lp = checkpoint(rk4_step, t, y, lp, use_reentrant=False) if ckpt else rk4_step(t, y, lp)
    block.odefunc._num_evals.fill_(4 * steps)
    return y, lp


def point_cnf_train(flow, x, context, logpx, e=None):
    """SequentialFlow in the forward (training / NLL) direction (cnf.py:33-48): [MBN, CNF x k, MBN]."""
    from ..models.cnf import CNF
    if e is None:
        e = torch.randn_like(x)                                           # odefunc.py:127-128
    for layer in flow.chain:
        if isinstance(layer, CNF):
            layer.odefunc._e = e
            x, logpx = cnf_block_train(layer, x, context, logpx, e)
        else:
            x, logpx = layer(x, context, logpx, None, False)
    return x, logpx
