"""Host wrappers of the training-tier entries of libcaspr_hip.so (include/caspr_hip_train.h).

These are the gradient kernels torch.autograd supplies in the reference (train_utils.py:173).  Same rules as
ops.py: float32 GPU tensors, point-major rows, no CPU fallback.
"""
import torch

from . import lib as _lib
from .ops import _chk_f32, _chk_rows, _p, _stream, _workspace


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
    _lib.check(L.caspr_conv1x1_wgrad_f32(_p(dy), lddy, _p(x), ldx, _p(in_scale), _p(in_shift), int(in_relu), int(in_relu_from), B, P, cin, cout,
                                         _p(dw), _p(dbias), int(accumulate), _p(ws), ws.numel(), _stream()), "caspr_conv1x1_wgrad_f32")
    return dw


def gn_bwd(y, da, C, mean, rstd, gamma, beta, dgamma, dbeta, groups=16, relu=True, accumulate=False):
    """GroupNorm(+ReLU) backward in place: da (B,P,>=C) becomes the gradient w.r.t. the raw conv output y."""
    _chk_f32(mean, rstd, gamma, beta, dgamma, dbeta)
    ldy, ldd = _chk_rows(y), _chk_rows(da)
    B, P, _ = y.shape
    L = _lib.load()
    ws = _workspace(L.caspr_gn_bwd_ws_bytes(B, P, C, groups), y.device)
    _lib.check(L.caspr_gn_bwd_f32(_p(y), ldy, _p(da), ldd, B, P, C, groups, _p(mean), _p(rstd), _p(gamma), _p(beta), int(relu),
                                  _p(dgamma), _p(dbeta), int(accumulate), _p(ws), ws.numel(), _stream()), "caspr_gn_bwd_f32")
    return da
