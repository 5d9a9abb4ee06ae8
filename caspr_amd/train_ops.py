"""Host wrappers of the training-tier entries of libcaspr_hip.so (include/caspr_hip_train.h).

These are the gradient kernels torch.autograd supplies in the reference (train_utils.py:173).  Same rules as
ops.py: float32 GPU tensors, point-major rows, no CPU fallback.
"""
import torch

from . import lib as _lib
from . import ops
from .ops import _chk_f32, _chk_i32, _chk_rows, _p, _stream, _workspace


def gn_stats_train(y, C, gamma, beta, groups=16, eps=1e-5, want_max=False):
    """As ops.gn_stats, and also the moments: -> scale (B,C), shift (B,C), mean (B,G), rstd (B,G) [, pmax (B,C)]."""
    _chk_f32(gamma, beta)
    ldy = _chk_rows(y)
    B, P, _ = y.shape
    dev = y.device
    scale = torch.empty(B, C, device=dev, dtype=torch.float32)
    shift = torch.empty(B, C, device=dev, dtype=torch.float32)
    mean = torch.empty(B, groups, device=dev, dtype=torch.float32)
    rstd = torch.empty(B, groups, device=dev, dtype=torch.float32)
    pmax = torch.empty(B, C, device=dev, dtype=torch.float32) if want_max else None
    L = _lib.load()
    ws = _workspace(L.caspr_gn_ws_bytes(B, P, C, groups), dev)
    _lib.check(L.caspr_gn_stats_train_f32(_p(y), ldy, B, P, C, groups, _p(gamma), _p(beta), float(eps), _p(scale), _p(shift), _p(pmax),
                                          _p(mean), _p(rstd), _p(ws), ws.numel(), _stream()), "caspr_gn_stats_train_f32")
    return (scale, shift, mean, rstd, pmax) if want_max else (scale, shift, mean, rstd)


def conv1x1_wgrad(dy, x, cin, cout, dw, dbias=None, in_scale=None, in_shift=None, in_relu=False, in_relu_from=0, accumulate=False):
    """dw (cout,cin) (+)= dy^T . in(x), dbias (cout) (+)= column sums of dy.  dy (B,P,>=cout), x (B,P,>=cin)."""
    _chk_f32(dw, dbias, in_scale, in_shift)
    lddy, ldx = _chk_rows(dy), _chk_rows(x)
    B, P, _ = x.shape
    if dy.shape[0] != B or dy.shape[1] != P:
        raise ValueError("conv1x1_wgrad: dy and x disagree on (B,P)")
    if tuple(dw.shape) != (cout, cin) or (dbias is not None and dbias.numel() != cout):
        raise ValueError("conv1x1_wgrad: dw must be (%d,%d)" % (cout, cin))
    L = _lib.load()
    ws = _workspace(L.caspr_wgrad_ws_bytes(B * P, cin, cout), x.device)
    # matrix products as ops.set_matmul_mode says: the exact bf16 three-way split where the tile is reasonably filled, f32 MFMA otherwise
    x6 = ops.CONV_BF16X6 and cin >= 32 and cout >= 32
    fn, name = (L.caspr_conv1x1_wgrad_bf16x6_f32, "caspr_conv1x1_wgrad_bf16x6_f32") if x6 else (L.caspr_conv1x1_wgrad_f32, "caspr_conv1x1_wgrad_f32")
    with ops.timed("k:%s:%d:%d:%d" % ("wgrad_bf16x6" if x6 else "wgrad_f32", cin, cout, B * P), 2):
        _lib.check(fn(_p(dy), lddy, _p(x), ldx, _p(in_scale), _p(in_shift), int(in_relu), int(in_relu_from), B, P, cin, cout,
                      _p(dw), _p(dbias), int(accumulate), _p(ws), ws.numel(), _stream()), name)
    return dw


def gn_bwd(y, da, C, mean, rstd, gamma, beta, dgamma, dbeta, groups=16, relu=True, accumulate=False, dmax=None, amax=None, out=None):
    """GroupNorm(+ReLU) backward.  da (B,P,>=C) or None (zero); dmax/amax (B,C) = gradient / arg-max point of the max over
    points of the un-rectified normalised feature.  Result in `out` (default: in place over da)."""
    _chk_f32(mean, rstd, gamma, beta, dgamma, dbeta, dmax)
    ldy = _chk_rows(y)
    B, P, _ = y.shape
    if out is None:
        if da is None:
            raise ValueError("gn_bwd: give `out` when da is None")
        out = da
    ldd, ldo = _chk_rows(da), _chk_rows(out)
    if amax is not None:
        _chk_i32(amax)
    L = _lib.load()
    ws = _workspace(L.caspr_gn_bwd_ws_bytes(B, P, C, groups), y.device)
    _lib.check(L.caspr_gn_bwd_f32(_p(y), ldy, _p(da), ldd, _p(dmax), _p(amax), _p(out), ldo, B, P, C, groups, _p(mean), _p(rstd), _p(gamma),
                                  _p(beta), int(relu), _p(dgamma), _p(dbeta), int(accumulate), _p(ws), ws.numel(), _stream()), "caspr_gn_bwd_f32")
    return out


def argmax_points(y, C, scale, shift):
    """(B,C) int32: first point attaining max_p (y*scale+shift)."""
    _chk_f32(scale, shift)
    ldy = _chk_rows(y)
    B, P, _ = y.shape
    out = torch.empty(B, C, device=y.device, dtype=torch.int32)
    L = _lib.load()
    ws = _workspace(L.caspr_argmax_ws_bytes(B, P, C), y.device)
    _lib.check(L.caspr_argmax_points_f32(_p(y), ldy, B, P, C, _p(scale), _p(shift), _p(out), _p(ws), ws.numel(), _stream()), "caspr_argmax_points_f32")
    return out


def colsum_batched(a, C):
    """(B,P,>=C) -> (B,C) sums over points."""
    ld = _chk_rows(a)
    B, P, _ = a.shape
    out = torch.empty(B, C, device=a.device, dtype=torch.float32)
    L = _lib.load()
    ws = _workspace(L.caspr_colsum_ws_bytes(B, P, C), a.device)
    _lib.check(L.caspr_colsum_batched_f32(_p(a), ld, B, P, C, _p(out), _p(ws), ws.numel(), _stream()), "caspr_colsum_batched_f32")
    return out


def three_interp_bwd(dout, idx, weight, C, dfeat):
    """dfeat (B,m,>=C) += scatter of dout (B,n,>=C) through the three-NN indices / weights."""
    _chk_f32(weight)
    _chk_i32(idx)
    ldo, ldf = _chk_rows(dout), _chk_rows(dfeat)
    B, n, _ = dout.shape
    m = dfeat.shape[1]
    _lib.check(_lib.load().caspr_three_interp_bwd_f32(_p(dout), ldo, _p(idx), _p(weight), B, m, n, C, _p(dfeat), ldf, _stream()),
               "caspr_three_interp_bwd_f32")
    return dfeat


def group_rows(xyz, new_xyz, feat, C, idx, centred=False, feat_kind=0, align=4):
    """-> G (B, M*ns, roundup(3+C, align)) rows [dxyz | feat | 0]; centred: minus the neighbourhood's sample-0 row
    (include/caspr_hip_train.h), feat_kind as ops.sa_mlp_max."""
    _chk_f32(xyz, new_xyz)
    _chk_i32(idx)
    B, n, _ = xyz.shape
    M, ns = idx.shape[1], idx.shape[2]
    ldf = 0 if feat is None else _chk_rows(feat)
    ldg = (3 + C + align - 1) // align * align
    G = torch.empty(B, M * ns, ldg, device=xyz.device, dtype=torch.float32)
    _lib.check(_lib.load().caspr_group_rows_f32(_p(xyz), _p(new_xyz), _p(feat), ldf, _p(idx), B, n, M, C, ns, int(bool(centred)), int(feat_kind),
                                                _p(G), ldg, _stream()),
               "caspr_group_rows_f32")
    return G


def group_rows_bwd(dG, idx, C, dfeat):
    """dfeat (B,n,>=C) += scatter of the feature columns of dG (B, M*ns, >=3+C)."""
    _chk_i32(idx)
    ldg, ldf = _chk_rows(dG), _chk_rows(dfeat)
    B, n, _ = dfeat.shape
    M, ns = idx.shape[1], idx.shape[2]
    _lib.check(_lib.load().caspr_group_rows_bwd_f32(_p(dG), ldg, _p(idx), B, n, M, C, ns, _p(dfeat), ldf, _stream()), "caspr_group_rows_bwd_f32")
    return dfeat


def gn_rows(y, ns, C, gamma, beta, relu, eps=1e-5, maxout=None):
    """Per-neighbourhood GroupNorm(16): y (B, M*ns, >=C).  -> (A (same shape, C cols) | None, mean, rstd, arg | None).
    maxout: (B, M, >=C) column slice receiving max over the ns rows (last layer of the point MLP)."""
    _chk_f32(gamma, beta)
    ldy = _chk_rows(y)
    NB = y.shape[0] * y.shape[1] // ns
    dev = y.device
    mean = torch.empty(NB, 16, device=dev, dtype=torch.float32)
    rstd = torch.empty(NB, 16, device=dev, dtype=torch.float32)
    A, arg, lda, ldm = None, None, 0, 0
    if maxout is None:
        A = torch.empty(y.shape[0], y.shape[1], C, device=dev, dtype=torch.float32)
        lda = C
    else:
        ldm = _chk_rows(maxout)
        arg = torch.empty(NB, C, device=dev, dtype=torch.int32)
    _lib.check(_lib.load().caspr_gn_rows_f32(_p(y), ldy, NB, ns, C, _p(gamma), _p(beta), float(eps), int(relu), _p(A), lda, _p(mean), _p(rstd),
                                             _p(maxout), ldm, _p(arg), _stream()), "caspr_gn_rows_f32")
    return A, mean, rstd, arg


def gn_rows_bwd(y, ns, C, gamma, beta, relu, mean, rstd, dgamma, dbeta, da=None, dmax=None, arg=None, out=None, accumulate=False):
    """Backward of gn_rows: da (dense) or dmax (B,M,>=C slice) + arg.  -> dY (same shape as y rows, C cols)."""
    _chk_f32(gamma, beta, mean, rstd, dgamma, dbeta)
    ldy = _chk_rows(y)
    NB = y.shape[0] * y.shape[1] // ns
    if out is None:
        out = da if da is not None else torch.empty(y.shape[0], y.shape[1], C, device=y.device, dtype=torch.float32)
    lda, ldm, ldo = _chk_rows(da), _chk_rows(dmax), _chk_rows(out)
    L = _lib.load()
    ws = _workspace(L.caspr_gn_rows_bwd_ws_bytes(C), y.device)
    _lib.check(L.caspr_gn_rows_bwd_f32(_p(y), ldy, NB, ns, C, _p(gamma), _p(beta), int(relu), _p(mean), _p(rstd), _p(da), lda, _p(dmax), ldm,
                                       _p(arg), _p(out), ldo, _p(dgamma), _p(dbeta), int(accumulate), _p(ws), ws.numel(), _stream()),
               "caspr_gn_rows_bwd_f32")
    return out


class Segments:
    """CSR of a scatter: for every target row the contributing source rows (ascending) and their weights.
    Built from flat int tensors `target` (nnz), optional `weight` (nnz) and optional `src_rows` (nnz; default: entry e
    reads source row e)."""

    def __init__(self, target, n_targets, weight=None, src_rows=None):
        tgt = target.reshape(-1).long()
        order = torch.sort(tgt, stable=True)[1]            # stable: equal targets keep ascending source order
        counts = torch.bincount(tgt, minlength=n_targets)
        self.start = torch.zeros(n_targets + 1, device=tgt.device, dtype=torch.int32)
        self.start[1:] = torch.cumsum(counts, 0).to(torch.int32)
        self.row = (order if src_rows is None else src_rows.reshape(-1)[order]).to(torch.int32).contiguous()
        self.w = None if weight is None else weight.reshape(-1)[order].contiguous().float()
        self.n_targets = n_targets


def segment_sum(src, seg, C, dst, col0=0, accumulate=True):
    """dst (targets, >=C) (+)= gather-sum of src rows (any leading shape, rows flattened) through `seg`, columns col0..col0+C."""
    src2 = src.reshape(-1, src.shape[-1])
    dst2 = dst.reshape(-1, dst.shape[-1])
    if src2.stride(1) != 1 or dst2.stride(1) != 1:
        raise ValueError("segment_sum: unit column stride required")
    _lib.check(_lib.load().caspr_segment_sum_f32(_p(src2), src2.stride(0), col0, _p(seg.start), _p(seg.row), _p(seg.w), seg.n_targets, C,
                                                 _p(dst2), dst2.stride(0), int(accumulate), _stream()), "caspr_segment_sum_f32")
    return dst
