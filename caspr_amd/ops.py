"""Python front-end of the C-ABI kernels (include/caspr_hip.h) on torch tensors.

Mirrors the operator surface the reference imports from Kaolin at models/pointnet2.py:7
(`furthest_point_sampling`, `fps_gather_by_index`, `ball_query`, `three_nn`, `three_interpolate`)
plus the fused / MFMA kernels of this build.  PyTorch only owns memory and streams here; every
tensor must live on a HIP device -- there is no CPU fallback (calls raise instead).

Layouts: activations are point-major (B, P, C) float32 with row stride = last-dim size unless a
leading-dimension is given; index tensors are int32.
"""
import ctypes

import torch

from . import lib as _lib
from .config import config as _cfg


# Optional stage timing with HIP events recorded on the launch stream (bench.py's roofline numbers).
TIMING = False
TIMERS = {}
TIMING_ONLY = None      # a set of names: at TIMING == True only these level-1 timers record (bench.py keeps the dominant kernel's
                        # pair inside the timed region and takes the stage breakdown from its separate detail pass)


class timed:
    """with ops.timed("name"): ...  -> appends a (start, end) event pair to TIMERS[name] when TIMING is on.
    level 2 = per-kernel timers (one pair per launch, keyed by kernel and shape): recorded only when TIMING >= 2, i.e. in
    bench.py's separate detail pass, never inside the timed region of the headline number."""

    def __init__(self, name, level=1):
        self.name = name
        self.level = level

    def __enter__(self):
        self.on = TIMING >= self.level and (TIMING_ONLY is None or TIMING != True or self.name in TIMING_ONLY)
        if self.on:
            self.t0 = torch.cuda.Event(enable_timing=True)
            self.t0.record(torch.cuda.current_stream())
        return self

    def __exit__(self, *exc):
        if self.on:
            t1 = torch.cuda.Event(enable_timing=True)
            t1.record(torch.cuda.current_stream())
            TIMERS.setdefault(self.name, []).append((self.t0, t1))
        return False


class untimed:
    """with ops.untimed(): ...  -> no timer records inside (the accuracy guard's check solves must not enter bench.py's per-kernel means)."""

    def __enter__(self):
        global TIMING
        self.prev, TIMING = TIMING, False
        return self

    def __exit__(self, *exc):
        global TIMING
        TIMING = self.prev
        return False


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _on_current_device(t):
    """The C entries launch on the CURRENT device's stream and never call hipSetDevice (include/caspr_hip.h): a tensor of
    another GPU would be dereferenced by the wrong device.  One process per GPU sets its device once (bench.py)."""
    if t.is_cuda and t.device.index != torch.cuda.current_device():
        raise ValueError("tensor lives on %s but the current device is cuda:%d: wrap the call in torch.cuda.device(t.device)"
                         % (t.device, torch.cuda.current_device()))


def _p(t):
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


def _chk_f32(*ts):
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise ValueError("caspr_amd kernels need tensors on the GPU (got %s): there is no CPU fallback" % t.device)
        _on_current_device(t)
        if t.dtype != torch.float32:
            raise TypeError("expected float32, got %s" % t.dtype)
        if not t.is_contiguous():
            raise ValueError("expected a contiguous tensor")


def _chk_rows(t):
    """(B,P,C) float32 GPU tensor whose rows may be a column slice of a wider buffer: returns the row stride."""
    if t is None:
        return 0
    if not t.is_cuda:
        raise ValueError("caspr_amd kernels need tensors on the GPU (got %s): there is no CPU fallback" % t.device)
    _on_current_device(t)
    if t.dtype != torch.float32 or t.dim() != 3 or t.stride(2) != 1 or t.stride(0) != t.shape[1] * t.stride(1):
        raise ValueError("expected a float32 (B,P,C) tensor with unit channel stride and packed rows")
    if t.stride(1) % 4 != 0 or t.data_ptr() % 16 != 0:
        raise ValueError("row stride must be a multiple of 4 floats and the base 16-byte aligned")
    return t.stride(1)


def _chk_i32(*ts):
    for t in ts:
        if not t.is_cuda or t.dtype != torch.int32 or not t.is_contiguous():
            raise ValueError("expected a contiguous int32 GPU tensor")


# ---------------------------------------------------------------------------------------------
def prep_input(x, quad=True, pairs=True):
    """x (B,T,N,4) -> xyz (B*T,N,3), feat (B*T,N,8) [x2,y2,z2,xz,xy,yz,0,0]  (tpointnet2.py:79-90)."""
    _chk_f32(x)
    B, T, N, _ = x.shape
    xyz = torch.empty(B * T, N, 3, device=x.device, dtype=torch.float32)
    feat = torch.empty(B * T, N, 8, device=x.device, dtype=torch.float32)
    _lib.check(_lib.load().caspr_prep_input_f32(_p(x), B * T, N, int(quad), int(pairs), _p(xyz), _p(feat), _stream()),
               "caspr_prep_input_f32")
    return xyz, feat


def furthest_point_sampling(xyz, M, guard=True, return_xyz=False):
    """Kaolin `furthest_point_sampling(xyz, M)` (pointnet2.py:384): (B,n,3) -> (B,M) int32."""
    _chk_f32(xyz)
    B, n, _ = xyz.shape
    idx = torch.empty(B, M, device=xyz.device, dtype=torch.int32)
    new_xyz = torch.empty(B, M, 3, device=xyz.device, dtype=torch.float32) if return_xyz else None
    _lib.check(_lib.load().caspr_fps_f32(_p(xyz), B, n, M, int(bool(guard)), _p(idx), _p(new_xyz), _stream()), "caspr_fps_f32")
    return (idx, new_xyz) if return_xyz else idx


def ball_query_pair(radius_a, ns_a, radius_b, ns_b, xyz, new_xyz):
    """The two ball queries of a set-abstraction level (one grouper per radius over the same xyz / new_xyz, pointnet2.py:338-342,391) in ONE
    pass over the cloud -> (idx_a (B,M,ns_a), idx_b (B,M,ns_b)), each what ball_query returns for its radius, bit for bit."""
    _chk_f32(xyz, new_xyz)
    B, n, _ = xyz.shape
    M = new_xyz.shape[1]
    ia = torch.empty(B, M, ns_a, device=xyz.device, dtype=torch.int32)
    ib = torch.empty(B, M, ns_b, device=xyz.device, dtype=torch.int32)
    _lib.check(_lib.load().caspr_ball_query2_f32(_p(xyz), _p(new_xyz), B, n, M, float(radius_a), ns_a, _p(ia), float(radius_b), ns_b, _p(ib), _stream()),
               "caspr_ball_query2_f32")
    return ia, ib


def gather_points(feat, idx):
    """Point-major `fps_gather_by_index`: feat (B,n,C), idx (B,M) -> (B,M,C)  (pointnet2.py:385)."""
    _chk_f32(feat)
    _chk_i32(idx)
    B, n, C = feat.shape
    M = idx.shape[1]
    out = torch.empty(B, M, C, device=feat.device, dtype=torch.float32)
    _lib.check(_lib.load().caspr_gather_points_f32(_p(feat), C, _p(idx), B, n, M, C, _p(out), C, _stream()), "caspr_gather_points_f32")
    return out


def fps_gather_by_index(feat_cf, idx):
    """Reference-shaped `fps_gather_by_index`: feat (B,C,n) channels-first -> (B,C,M)."""
    return gather_points(feat_cf.transpose(1, 2).contiguous(), idx).transpose(1, 2)


def ball_query(radius, ns, xyz, new_xyz):
    """Kaolin `ball_query(radius, ns, xyz, new_xyz)` (pointnet2.py:340-342,391) -> (B,M,ns) int32."""
    _chk_f32(xyz, new_xyz)
    B, n, _ = xyz.shape
    M = new_xyz.shape[1]
    idx = torch.empty(B, M, ns, device=xyz.device, dtype=torch.int32)
    _lib.check(_lib.load().caspr_ball_query_f32(_p(xyz), _p(new_xyz), B, n, M, float(radius), ns, _p(idx), _stream()), "caspr_ball_query_f32")
    return idx


def group_points(xyz, new_xyz, feat, idx):
    """Reference-shaped grouper output (B,M,3+C,ns) (pointnet2.py:391-398); feat point-major (B,n,C) or None."""
    _chk_f32(xyz, new_xyz, feat)
    _chk_i32(idx)
    B, n, _ = xyz.shape
    M, ns = idx.shape[1], idx.shape[2]
    C = 0 if feat is None else feat.shape[2]
    out = torch.empty(B, M, 3 + C, ns, device=xyz.device, dtype=torch.float32)
    _lib.check(_lib.load().caspr_group_points_f32(_p(xyz), _p(new_xyz), _p(feat), C, _p(idx), B, n, M, C, ns, _p(out), _stream()),
               "caspr_group_points_f32")
    return out


def three_nn(unknown, known, with_weights=False):
    """Kaolin `three_nn` (pointnet2.py:514): -> dist (B,n,3) [sqrt], idx (B,n,3) [, normalised inverse-distance weights :516-518]."""
    _chk_f32(unknown, known)
    B, n, _ = unknown.shape
    m = known.shape[1]
    dist = torch.empty(B, n, 3, device=unknown.device, dtype=torch.float32)
    idx = torch.empty(B, n, 3, device=unknown.device, dtype=torch.int32)
    w = torch.empty(B, n, 3, device=unknown.device, dtype=torch.float32) if with_weights else None
    _lib.check(_lib.load().caspr_three_nn_f32(_p(unknown), _p(known), B, n, m, _p(dist), _p(idx), _p(w), _stream()), "caspr_three_nn_f32")
    return (dist, idx, w) if with_weights else (dist, idx)


def three_interpolate(feat, idx, weight, skip=None, skip_channels=None, in_scale=None, in_shift=None, in_relu=False, C=None, align=4):
    """Point-major `three_interpolate` (+ concat of skip features): feat (B,m,>=C) -> (B,n,roundup(C+C2, align))
    (pointnet2.py:519-523), columns past C+C2 zero.  in_scale/in_shift (B,C): feat is read as relu(feat*scale+shift)."""
    _chk_f32(weight, in_scale, in_shift)
    _chk_i32(idx)
    ldf = _chk_rows(feat)
    lds = _chk_rows(skip)
    B, m, _ = feat.shape
    C = feat.shape[2] if C is None else C
    n = idx.shape[1]
    C2 = 0 if skip is None else (skip.shape[2] if skip_channels is None else skip_channels)
    if align % 4:
        raise ValueError("three_interpolate: align must be a multiple of 4")
    ldo = (C + C2 + align - 1) // align * align
    out = torch.empty(B, n, ldo, device=feat.device, dtype=torch.float32)
    _lib.check(_lib.load().caspr_three_interp_f32(_p(feat), ldf, _p(idx), _p(weight), _p(in_scale), _p(in_shift), int(in_relu),
                                                  _p(skip), lds, B, m, n, C, C2, _p(out), ldo, _stream()),
               "caspr_three_interp_f32")
    return out


def three_interp_add_gn(u, idx, weight, skip, skip_channels, wskip, bias, gamma, beta, groups=16, eps=1e-5):
    """Feature propagation's first layer AFTER its conv ran on the coarse level (include/caspr_hip.h: caspr_three_interp_add_gn_f32):
    u (B,m,>=C) = W_p h, idx / weight (B,n,3) from three_nn, skip (B,n,>=C2) | None, wskip (C,C2) -> (y (B,n,C) raw, scale (B,C), shift (B,C))."""
    _chk_f32(u, weight, skip, wskip, bias, gamma, beta)
    _chk_i32(idx)
    ldu, lds = _chk_rows(u), _chk_rows(skip)
    B, m, _ = u.shape
    n = idx.shape[1]
    C = gamma.numel()
    C2 = 0 if skip is None else int(skip_channels)
    y = torch.empty(B, n, C, device=u.device, dtype=torch.float32)
    scale = torch.empty(B, C, device=u.device, dtype=torch.float32)
    shift = torch.empty(B, C, device=u.device, dtype=torch.float32)
    L = _lib.load()
    nb = L.caspr_three_interp_add_gn_ws_bytes(B, n, groups)
    ws = _workspace(nb, u.device)
    with timed("k:three_interp_add_gn:%d:%d:%d" % (C2, C, B * n), 2):
        _lib.check(L.caspr_three_interp_add_gn_f32(_p(u), ldu, _p(idx), _p(weight), _p(skip), lds, C2, _p(wskip), _p(bias), B, m, n, C, _p(y), C,
                                                   groups, _p(gamma), _p(beta), float(eps), _p(scale), _p(shift), _p(ws), ws.numel(), _stream()),
                   "caspr_three_interp_add_gn_f32")
    return y, scale, shift


# ---------------------------------------------------------------------------------------------
# Matrix products.  Every contraction of this model is f32 arithmetic; there are two ways to run it on gfx950:
#   "bf16x6" (default): each f32 operand is split EXACTLY into three bf16 numbers (x = x1 + x2 + x3) and a product is
#       evaluated as the six partial products a3b1 + a2b2 + a1b3 + a2b1 + a1b2 + a1b1 on the bf16 matrix pipe with f32
#       accumulation (csrc/gemm_bf16x6.hip, csrc/ode_bf16x6.hip): every partial product is exact, the dropped terms are
#       below 2^-23 |a||b| -- measured at or below the f32 MFMA kernels' own error against f64 on every test -- at 16x the
#       per-instruction FLOP rate, i.e. a 2.67x higher ceiling (2500 / 6 = 416.7 f32-equivalent TFLOP/s vs 157.3);
#   "f32": v_mfma_f32_16x16x4_f32 only (csrc/gemm.hip, csrc/ode.hip), bit-for-bit an ordered fmaf chain.
# Shapes the bf16x6 kernels do not cover (Cin < 192 or not a multiple of 32, fewer than 128 rows per batch entry, the
# set-abstraction MLPs, the latent ODE) run on the f32 MFMA kernels in either mode.  config.matmul = "f32" (or, under
# CASPR_DEBUG=1 only, CASPR_MATMUL=f32) selects the f32 kernels at import; set_matmul_mode() switches at run time (bench.py times both in one process).
_mode = _cfg.matmul            # caspr_amd/config.py (the environment only under CASPR_DEBUG=1)
CONV_BF16X6 = _mode == "bf16x6"      # pointwise convs (conv1x1) on the bf16x6 kernel where the shape allows
CONV_X6W = _cfg.conv_x6w             # ... and the layers with >= 512 output channels on the 512-channel kernel
CNF_BF16X6 = _mode == "bf16x6"       # point-CNF solves on the bf16x6 kernel
_X6_MIN_CIN = 192     # below this the f32 LDS kernel is used anyway (set-abstraction / input layers)
# the 512-channel kernel (gemm_bf16x6w.hip) from this many input channels.  Round 3: 1024 (one workgroup per tile, prologue / epilogue
# exposed: only the 1600-wide head layer's 50-chunk K loop amortised them).  Round 4: the kernel is persistent and its statistics
# epilogue shorter (6.8 vs 7.07 ms on the head layer), so the layers with >= 512 output channels, >= 512 input channels and >= 1024 rows
# per batch entry (_X6W_MIN_ROWS: the coarse levels -- 640 tiles at cfg-2 -- quantise badly onto a persistent grid of 256 workgroups) take it:
# in-step at cfg-2 576 -> 1600: 3.40 -> 3.24 ms, 512 -> 512: 0.92 -> 0.87, 544 -> 512: 0.89 -> 0.84; the 128 -> 1024 layer (4 k-chunks per
# tile) and the 81,920-row levels are faster on the 256-channel kernel.  (config.x6w_min_cin)
_X6W_MIN_CIN = _cfg.x6w_min_cin
_X6W_MIN_ROWS = 1024
_X6_GN_MIN_CIN = 64   # conv + GroupNorm statistics in one pass (conv1x1_gn): pays from a smaller width (no second pass over the output)


def _x6w_fills(pw, B, P):
    """Rows per batch entry from which the persistent 512-channel kernel is chosen.  By the entry's shape only, NEVER by the number of
    entries: a sequence's result must not depend on the batch around it (bitwise sharding invariance), and the two kernels round
    differently."""
    return P >= _X6W_MIN_ROWS


def set_matmul_mode(mode=None, conv=None, cnf=None):
    """Select the kernels of the matrix products: mode "bf16x6" | "f32", or conv= / cnf= booleans individually.
    Returns the previous (conv, cnf) pair.  Packed weights of both kinds are built on first use and cached."""
    global CONV_BF16X6, CNF_BF16X6
    prev = (CONV_BF16X6, CNF_BF16X6)
    if mode is not None:
        if mode not in ("bf16x6", "f32"):
            raise ValueError("mode must be 'bf16x6' or 'f32'")
        CONV_BF16X6 = CNF_BF16X6 = mode == "bf16x6"
    if conv is not None:
        CONV_BF16X6 = bool(conv)
    if cnf is not None:
        CNF_BF16X6 = bool(cnf)
    return prev


def matmul_mode():
    return {"conv": "bf16x6" if CONV_BF16X6 else "f32", "cnf": "bf16x6" if CNF_BF16X6 else "f32"}


class PackedWeight:
    """A (Cout, Cin) weight matrix in MFMA A-fragment order (see csrc/common.h), plus -- built on first use -- the
    three-way bf16 split of csrc/gemm_bf16x6.hip when the shape is supported (`x3()`)."""

    def __init__(self, w2d, col0=0, ncols=None):
        _chk_f32(w2d)
        self.cout, ldw = w2d.shape
        self.cin = ldw - col0 if ncols is None else ncols
        size = _lib.load().caspr_packed_size(self.cout, self.cin)
        self.data = torch.empty(size, device=w2d.device, dtype=torch.float32)
        _lib.check(_lib.load().caspr_pack_weight_f32(_p(w2d), ldw, self.cout, col0, self.cin, _p(self.data), _stream()),
                   "caspr_pack_weight_f32")
        self.x6_ok = self.cin % 32 == 0 and self.cin >= _X6_MIN_CIN and self.cout % 4 == 0 and self.cout >= 128
        self.x6_gn_ok = self.cin % 32 == 0 and self.cin >= _X6_GN_MIN_CIN and self.cout % 4 == 0 and self.cout >= 128
        self._x3 = None
        self._xw = None
        # the 128-point x 512-channel kernel (csrc/gemm_bf16x6w.hip) takes the layers with >= 512 output channels and >= 8 k chunks
        self.x6w_ok = self.cin % 32 == 0 and self.cin >= _X6W_MIN_CIN and self.cout % 4 == 0 and self.cout >= 512
        # kept for the lazy bf16x3 pack only, with the version the f32 pack was taken at: w2d may be a VIEW of a parameter
        # (detach()[:, :, 0]), and a pack made after an in-place update would silently disagree with self.data
        self._src = (w2d, ldw, col0, w2d._version) if (self.x6_gn_ok or self.x6w_ok) else None

    def x3(self):
        if self._x3 is None:
            w2d, ldw, col0, version = self._src
            if w2d._version != version:
                raise RuntimeError("PackedWeight.x3(): the weight tensor was modified in place after this pack was built; re-create the "
                                   "PackedWeight (WeightCache does: it is keyed by (data_ptr, _version))")
            nbytes = _lib.load().caspr_bf16x3_packed_bytes(self.cout, self.cin)
            self._x3 = torch.empty(nbytes, device=w2d.device, dtype=torch.uint8)
            _lib.check(_lib.load().caspr_pack_weight_bf16x3(_p(w2d), ldw, self.cout, col0, self.cin, _p(self._x3), _stream()),
                       "caspr_pack_weight_bf16x3")
            if not self.x6w_ok or self._xw is not None:
                self._src = None        # every pack exists: release the (possibly padded) copy of the weight
        return self._x3

    def xw(self):
        """(main, tail) packs of caspr_conv1x1_x6w_f32: the first cout - cout % 512 rows for the 512-channel kernel, the rest (or
        None) as a bf16x3 pack for the 256-channel kernel."""
        if self._xw is None:
            w2d, ldw, col0, version = self._src
            if w2d._version != version:
                raise RuntimeError("PackedWeight.xw(): the weight tensor was modified in place after this pack was built")
            L = _lib.load()
            cmain = self.cout - self.cout % 512
            main = torch.empty(L.caspr_x6w_packed_bytes(cmain, self.cin), device=w2d.device, dtype=torch.uint8)
            _lib.check(L.caspr_pack_weight_x6w(_p(w2d), ldw, cmain, col0, self.cin, _p(main), _stream()), "caspr_pack_weight_x6w")
            tail = None
            if cmain != self.cout:
                wt = w2d[cmain:]
                tail = torch.empty(L.caspr_bf16x3_packed_bytes(self.cout - cmain, self.cin), device=w2d.device, dtype=torch.uint8)
                _lib.check(L.caspr_pack_weight_bf16x3(_p(wt), ldw, self.cout - cmain, col0, self.cin, _p(tail), _stream()), "caspr_pack_weight_bf16x3")
            self._xw = (main, tail)
            if self._x3 is not None:
                self._src = None
        return self._xw


CONV_ROW_INVARIANT = 0x100     # include/caspr_hip.h: act flag


def conv1x1(pw, bias, x, bbias=None, in_scale=None, in_shift=None, in_relu=False, in_relu_from=0, act=0, out=None, row_invariant=False):
    """Pointwise conv on MFMA: x (B,P,>=Cin) point-major (may be a column slice of a wider buffer) ->
    (B,P,roundup4(Cout)) or into `out` (same rules).  See caspr_conv1x1_f32.
    row_invariant: a row's result depends on that row only (not on P / its position): for convs over frames-as-rows."""
    if row_invariant:
        act = act | CONV_ROW_INVARIANT
    _chk_f32(bias, bbias, in_scale, in_shift)
    ldx = _chk_rows(x)
    B, P, _ = x.shape
    if x.shape[2] < pw.cin and ldx < (pw.cin + 3) // 4 * 4:
        raise ValueError("conv1x1: input rows hold %d channels, weight needs %d" % (x.shape[2], pw.cin))
    if out is None:
        out = torch.empty(B, P, (pw.cout + 3) // 4 * 4, device=x.device, dtype=torch.float32)
    ldy = _chk_rows(out)
    if CONV_BF16X6 and CONV_X6W and pw.x6w_ok and P % 128 == 0 and not row_invariant and in_relu_from % 8 == 0 and act == 0 and _x6w_fills(pw, B, P):
        main, tail = pw.xw()
        with timed("k:conv1x1_bf16x6:%d:%d:%d" % (pw.cin, pw.cout, B * P), 2):
            _lib.check(_lib.load().caspr_conv1x1_x6w_f32(_p(main), _p(tail), _p(bias), _p(bbias), _p(x), ldx, _p(in_scale), _p(in_shift), int(in_relu),
                                                         int(in_relu_from), _p(out), ldy, B, P, pw.cin, pw.cout, 0, None, None, 0.0, None, None, None,
                                                         None, None, None, 0, _stream()), "caspr_conv1x1_x6w_f32")
        return out
    if CONV_BF16X6 and pw.x6_ok and P % 128 == 0 and not row_invariant and in_relu_from % 8 == 0:
        with timed("k:conv1x1_bf16x6:%d:%d:%d" % (pw.cin, pw.cout, B * P), 2):
            _lib.check(_lib.load().caspr_conv1x1_bf16x6_f32(_p(pw.x3()), _p(bias), _p(bbias), _p(x), ldx, _p(in_scale), _p(in_shift), int(in_relu),
                                                            int(in_relu_from), _p(out), ldy, B, P, pw.cin, pw.cout, act, _stream()),
                       "caspr_conv1x1_bf16x6_f32")
        return out
    with timed("k:conv1x1_f32:%d:%d:%d" % (pw.cin, pw.cout, B * P), 2):
        _lib.check(_lib.load().caspr_conv1x1_f32(_p(pw.data), _p(bias), _p(bbias), _p(x), ldx, _p(in_scale), _p(in_shift), int(in_relu),
                                                 int(in_relu_from), _p(out), ldy, B, P, pw.cin, pw.cout, act, _stream()), "caspr_conv1x1_f32")
    return out


_ws_cache = {}


def _workspace(nbytes, device):
    key = (device.index, torch.cuda.current_stream().cuda_stream)
    ws = _ws_cache.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(max(nbytes, 1 << 20), device=device, dtype=torch.uint8)
        _ws_cache[key] = ws
    return ws


def gn_stats(y, C, gamma, beta, groups=16, eps=1e-5, want_max=False):
    """GroupNorm statistics of y (B,P,ldy) -> scale (B,C), shift (B,C) [, max over points of the normalised output (B,C)]."""
    _chk_f32(gamma, beta)
    ldy = _chk_rows(y)
    B, P, _ = y.shape
    scale = torch.empty(B, C, device=y.device, dtype=torch.float32)
    shift = torch.empty(B, C, device=y.device, dtype=torch.float32)
    pmax = torch.empty(B, C, device=y.device, dtype=torch.float32) if want_max else None
    nbytes = _lib.load().caspr_gn_ws_bytes(B, P, C, groups)
    ws = _workspace(nbytes, y.device)
    _lib.check(_lib.load().caspr_gn_stats_f32(_p(y), ldy, B, P, C, groups, _p(gamma), _p(beta), float(eps), _p(scale), _p(shift), _p(pmax),
                                              _p(ws), ws.numel(), _stream()), "caspr_gn_stats_f32")
    return (scale, shift, pmax) if want_max else (scale, shift)


def conv1x1_gn(pw, bias, x, gamma, beta, groups=16, eps=1e-5, want_max=False, want_moments=False, write=True, bbias=None,
               in_scale=None, in_shift=None, in_relu=False, in_relu_from=0, out=None, pool=1):
    """conv1x1 followed by the statistics of the GroupNorm on its output (the model's conv -> GroupNorm -> ReLU block):
    -> (y | None, scale (B,C), shift (B,C)[, mean (B,G), rstd (B,G)][, pmax (B,C)]).  On the bf16x6 path the statistics come
    out of the conv's epilogue (caspr_conv1x1_gn_bf16x6_f32) and `write=False` skips the output altogether; otherwise this is
    conv1x1 + gn_stats (`write` is then ignored: y is returned)."""
    _chk_f32(bias, bbias, in_scale, in_shift, gamma, beta)
    B, P, _ = x.shape
    C = pw.cout
    # pool: the statistics are taken over `pool` consecutive batch entries together (-> (B / pool, C) outputs) while in_scale /
    # in_shift / bbias stay per entry (caspr_conv1x1_gn_pooled_bf16x6_f32)
    if pool < 1 or B % pool:
        raise ValueError("conv1x1_gn: pool=%d must divide the %d batch entries" % (pool, B))
    if not (CONV_BF16X6 and pw.x6_gn_ok and P % 128 == 0 and C % groups == 0 and in_relu_from % 8 == 0):
        y = conv1x1(pw, bias, x, bbias=bbias, in_scale=in_scale, in_shift=in_shift, in_relu=in_relu, in_relu_from=in_relu_from, out=out)
        yg = y.view(B // pool, pool * P, y.shape[2]) if pool > 1 else y
        if want_moments:
            from . import train_ops
            return (y,) + tuple(train_ops.gn_stats_train(yg, C, gamma, beta, groups, eps, want_max))
        return (y,) + tuple(gn_stats(yg, C, gamma, beta, groups, eps, want_max))
    ldx = _chk_rows(x)
    if x.shape[2] < pw.cin and ldx < (pw.cin + 3) // 4 * 4:
        raise ValueError("conv1x1_gn: input rows hold %d channels, weight needs %d" % (x.shape[2], pw.cin))
    dev = x.device
    y = None
    if write or out is not None:
        y = out if out is not None else torch.empty(B, P, (C + 3) // 4 * 4, device=dev, dtype=torch.float32)
    ldy = _chk_rows(y) if y is not None else 0
    Bs = B // pool
    scale = torch.empty(Bs, C, device=dev, dtype=torch.float32)
    shift = torch.empty(Bs, C, device=dev, dtype=torch.float32)
    mean = torch.empty(Bs, groups, device=dev, dtype=torch.float32) if want_moments else None
    rstd = torch.empty(Bs, groups, device=dev, dtype=torch.float32) if want_moments else None
    pmax = torch.empty(Bs, C, device=dev, dtype=torch.float32) if want_max else None
    L = _lib.load()
    ws = _workspace(L.caspr_conv_gn_ws_bytes(B, P, C), dev)
    if pool > 1 and CONV_X6W and pw.x6w_ok and _x6w_fills(pw, B, P):
        main, tail = pw.xw()
        with timed("k:conv1x1_bf16x6:%d:%d:%d" % (pw.cin, C, B * P), 2):
            _lib.check(L.caspr_conv1x1_x6w_pooled_f32(_p(main), _p(tail), _p(bias), _p(bbias), _p(x), ldx, _p(in_scale), _p(in_shift), int(in_relu),
                                                      int(in_relu_from), _p(y), ldy, B, P, pw.cin, C, groups, int(pool), _p(gamma), _p(beta), float(eps),
                                                      _p(scale), _p(shift), _p(pmax), _p(mean), _p(rstd), _p(ws), ws.numel(), _stream()),
                       "caspr_conv1x1_x6w_pooled_f32")
        res = (y, scale, shift)
        if want_moments:
            res += (mean, rstd)
        if want_max:
            res += (pmax,)
        return res
    if pool > 1:
        with timed("k:conv1x1_bf16x6:%d:%d:%d" % (pw.cin, C, B * P), 2):
            _lib.check(L.caspr_conv1x1_gn_pooled_bf16x6_f32(_p(pw.x3()), _p(bias), _p(bbias), _p(x), ldx, _p(in_scale), _p(in_shift), int(in_relu),
                                                            int(in_relu_from), _p(y), ldy, B, P, pw.cin, C, groups, int(pool), _p(gamma), _p(beta),
                                                            float(eps), _p(scale), _p(shift), _p(pmax), _p(mean), _p(rstd), _p(ws), ws.numel(),
                                                            _stream()), "caspr_conv1x1_gn_pooled_bf16x6_f32")
        res = (y, scale, shift)
        if want_moments:
            res += (mean, rstd)
        if want_max:
            res += (pmax,)
        return res
    if CONV_X6W and pw.x6w_ok and _x6w_fills(pw, B, P):
        main, tail = pw.xw()
        with timed("k:conv1x1_bf16x6:%d:%d:%d" % (pw.cin, C, B * P), 2):
            _lib.check(L.caspr_conv1x1_x6w_f32(_p(main), _p(tail), _p(bias), _p(bbias), _p(x), ldx, _p(in_scale), _p(in_shift), int(in_relu),
                                               int(in_relu_from), _p(y), ldy, B, P, pw.cin, C, groups, _p(gamma), _p(beta), float(eps),
                                               _p(scale), _p(shift), _p(pmax), _p(mean), _p(rstd), _p(ws), ws.numel(), _stream()),
                       "caspr_conv1x1_x6w_f32")
        res = (y, scale, shift)
        if want_moments:
            res += (mean, rstd)
        if want_max:
            res += (pmax,)
        return res
    with timed("k:conv1x1_bf16x6:%d:%d:%d" % (pw.cin, C, B * P), 2):     # conv + statistics epilogue + the finalize kernel
        _lib.check(L.caspr_conv1x1_gn_bf16x6_f32(_p(pw.x3()), _p(bias), _p(bbias), _p(x), ldx, _p(in_scale), _p(in_shift), int(in_relu),
                                                 int(in_relu_from), _p(y), ldy, B, P, pw.cin, C, groups, _p(gamma), _p(beta), float(eps),
                                                 _p(scale), _p(shift), _p(pmax), _p(mean), _p(rstd), _p(ws), ws.numel(), _stream()),
                   "caspr_conv1x1_gn_bf16x6_f32")
    res = (y, scale, shift)
    if want_moments:
        res += (mean, rstd)
    if want_max:
        res += (pmax,)
    return res


def conv1x1_gn_early_ok(pw, B, P, groups, early_channels):
    """Can conv1x1_gn_early run this layer in pieces?  The 512-channel kernel must take it, in at least two channel tiles, and the
    channels the caller wants early must lie inside GroupNorm group 0, inside the first tile."""
    C = pw.cout
    return bool(CONV_BF16X6 and CONV_X6W and pw.x6w_ok and P % 128 == 0 and _x6w_fills(pw, B, P) and C % groups == 0 and C // 512 >= 2
                and early_channels <= min(C // groups, 512))


def conv1x1_gn_early(pw, bias, x, gamma, beta, on_early, groups=16, eps=1e-5, in_scale=None, in_shift=None, in_relu=False, in_relu_from=0,
                     reserve_cus=1, tail_stream=None):
    """conv1x1_gn(..., want_max=True) in two pieces (caspr_conv1x1_x6w_part_f32 / caspr_conv_gn_finalize_f32): after the FIRST 512-channel
    tile the statistics of GroupNorm group 0 are final -- `on_early(pmax)` is called with the (B, C) max-over-points tensor whose
    group-0 columns are valid, on the current stream, and may queue work on another stream behind an event (the latent solve: it
    starts from pmax[:, :64], caspr.py:169) -- then the remaining tiles run on all but `reserve_cus` compute units, and the full
    finalize follows.  Returns (y, scale, shift, pmax) with the values of conv1x1_gn, bit for bit.
    tail_stream: the stream on_early queued its work on.  The < 512-channel remainder of the layer (a pass of its own kind over the whole
    input, bound by reading it once: 0.6 ms at cfg-2) is then queued THERE, behind that work, instead of behind the main tiles: the
    compute units reserved for the early kernel are free again long before the main tiles finish (the solve takes 2.6 of their 4.8 ms),
    and the remainder runs on them beside the tiles instead of after them."""
    _chk_f32(bias, in_scale, in_shift, gamma, beta)
    B, P, _ = x.shape
    C = pw.cout
    if not conv1x1_gn_early_ok(pw, B, P, groups, 1):
        raise ValueError("conv1x1_gn_early: this layer does not run on the 512-channel kernel in >= 2 channel tiles")
    ldx = _chk_rows(x)
    dev = x.device
    y = torch.empty(B, P, (C + 3) // 4 * 4, device=dev, dtype=torch.float32)
    ldy = _chk_rows(y)
    scale = torch.empty(B, C, device=dev, dtype=torch.float32)
    shift = torch.empty(B, C, device=dev, dtype=torch.float32)
    pmax = torch.empty(B, C, device=dev, dtype=torch.float32)
    L = _lib.load()
    ws = _workspace(L.caspr_conv_gn_ws_bytes(B, P, C), dev)
    main, tail = pw.xw()
    mt_all = C // 512

    def part(mt0, mt1, with_tail, reserve):
        _lib.check(L.caspr_conv1x1_x6w_part_f32(_p(main), _p(tail), _p(bias), None, _p(x), ldx, _p(in_scale), _p(in_shift), int(in_relu), int(in_relu_from),
                                                _p(y), ldy, B, P, pw.cin, C, mt0, mt1, int(with_tail), int(reserve), _p(ws), ws.numel(), _stream()),
                   "caspr_conv1x1_x6w_part_f32")

    def finalize(g0, g1):
        _lib.check(L.caspr_conv_gn_finalize_f32(_p(ws), ws.numel(), B, P, C, groups, g0, g1, 1, _p(gamma), _p(beta), float(eps), _p(scale), _p(shift),
                                                _p(pmax), None, None, _stream()), "caspr_conv_gn_finalize_f32")
    with timed("k:conv1x1_bf16x6:%d:%d:%d" % (pw.cin, C, B * P), 2):
        part(0, 1, False, 0)
        finalize(0, 1)
        on_early(pmax)
        if tail_stream is not None and tail is not None and reserve_cus > 0:
            main_stream = torch.cuda.current_stream()
            issued = torch.cuda.Event()
            issued.record(main_stream)               # the remainder reads x and writes its own columns of y / of the partials: nothing of tile 0's
            with torch.cuda.stream(tail_stream):
                tail_stream.wait_event(issued)
                part(mt_all, mt_all, True, 0)        # the remainder only, behind whatever on_early queued on that stream
                tail_done = torch.cuda.Event()
                tail_done.record(tail_stream)
            part(1, mt_all, False, reserve_cus)
            main_stream.wait_event(tail_done)
            for t_ in (x, y, ws, in_scale, in_shift):
                if t_ is not None:
                    t_.record_stream(tail_stream)
        else:
            part(1, mt_all, True, reserve_cus)
        finalize(1, groups)         # group 0 is final already (and the side stream may be reading its pmax / scale / shift right now)
    return y, scale, shift, pmax


def conv1x1_gn_tail_beside(pw, bias, x, gamma, beta, tail_stream, groups=16, eps=1e-5, bbias=None, in_scale=None, in_shift=None, in_relu=False,
                           in_relu_from=0, pool=1):
    """conv1x1_gn(...) of a layer that runs on the persistent 512-channel kernel with a < 512-channel remainder (1600 = 3 x 512 + 64), the
    remainder's pass -- bound by reading the input once -- queued on `tail_stream` BESIDE the main tiles instead of behind them (its
    workgroups need 4 KB of LDS and 116 registers: they share the compute units with the persistent kernel's one wave per SIMD).  Same
    pieces (caspr_conv1x1_x6w_part_f32 / caspr_conv_gn_finalize_f32), same bits as conv1x1_gn; falls back to it when the layer does not
    take that kernel.  -> (y, scale, shift)."""
    B, P, _ = x.shape
    C = pw.cout
    ok = (tail_stream is not None and CONV_BF16X6 and CONV_X6W and pw.x6w_ok and pw.x6_gn_ok and P % 128 == 0 and C % groups == 0 and in_relu_from % 8 == 0
          and C % 512 != 0 and C > 512 and B % pool == 0 and _x6w_fills(pw, B, P) and not torch.cuda.is_current_stream_capturing())
    if not ok:
        return conv1x1_gn(pw, bias, x, gamma, beta, groups, eps, bbias=bbias, in_scale=in_scale, in_shift=in_shift, in_relu=in_relu,
                          in_relu_from=in_relu_from, pool=pool)
    _chk_f32(bias, bbias, in_scale, in_shift, gamma, beta)
    ldx = _chk_rows(x)
    dev = x.device
    y = torch.empty(B, P, (C + 3) // 4 * 4, device=dev, dtype=torch.float32)
    ldy = _chk_rows(y)
    Bs = B // pool
    scale = torch.empty(Bs, C, device=dev, dtype=torch.float32)
    shift = torch.empty(Bs, C, device=dev, dtype=torch.float32)
    L = _lib.load()
    ws = _workspace(L.caspr_conv_gn_ws_bytes(B, P, C), dev)
    main, tail = pw.xw()
    mt_all = C // 512

    def part(mt0, mt1, with_tail):
        _lib.check(L.caspr_conv1x1_x6w_part_f32(_p(main), _p(tail), _p(bias), _p(bbias), _p(x), ldx, _p(in_scale), _p(in_shift), int(in_relu), int(in_relu_from),
                                                _p(y), ldy, B, P, pw.cin, C, mt0, mt1, int(with_tail), 0, _p(ws), ws.numel(), _stream()),
                   "caspr_conv1x1_x6w_part_f32")
    with timed("k:conv1x1_bf16x6:%d:%d:%d" % (pw.cin, C, B * P), 2):
        main_stream = torch.cuda.current_stream()
        tail_stream.wait_stream(main_stream)
        with torch.cuda.stream(tail_stream):
            part(mt_all, mt_all, True)               # the remainder only: its own columns of y and of the partials
            tail_done = torch.cuda.Event()
            tail_done.record(tail_stream)
        part(0, mt_all, False)
        main_stream.wait_event(tail_done)
        for t_ in (x, y, ws, in_scale, in_shift, bbias, bias):
            if t_ is not None:
                t_.record_stream(tail_stream)
        _lib.check(L.caspr_conv_gn_finalize_f32(_p(ws), ws.numel(), B, P, C, groups, 0, groups, int(pool), _p(gamma), _p(beta), float(eps), _p(scale), _p(shift),
                                                None, None, None, _stream()), "caspr_conv_gn_finalize_f32")
    return y, scale, shift


FEAT_QUAD, FEAT_PAIRS, FEAT_LO_IN, FEAT_LO_OUT = 1, 2, 4, 8     # include/caspr_hip.h
SA_ONLY_MFMA, SA_ONLY_F64 = 16, 32                              # the call in two halves (two streams): the MFMA kernel / the f64 re-evaluation


def sa_mlp_max(xyz, new_xyz, feat, idx, C, layers, out, out_off, feat_kind=0, part=None):
    """Fused grouper + 3-layer point MLP + GroupNorm + max (pointnet2.py:391-409,649-703).
    feat (B,n,ldf) point-major with C valid channels; layers = 3 x (PackedWeight, bias, gamma, beta).
    feat_kind: FEAT_QUAD | FEAT_PAIRS when feat is prep_input's quadratic augmentation of xyz (first level); FEAT_LO_OUT: `out` rows are
    [channels | their low parts] (the second half of the row receives what the f32 output lacks of the kernel's f64 result); FEAT_LO_IN:
    `feat` rows are [C channels .. | low parts from column ldf / 2] as a FEAT_LO_OUT call wrote them (include/caspr_hip.h).
    part: None = the whole call; "mfma" / "f64" = one of its two halves (SA_ONLY_MFMA / SA_ONLY_F64: the MFMA kernel over the neighbourhoods
    the f64 re-evaluation does not take / that re-evaluation alone), for a caller that runs them on two streams -- together they write what
    the whole call writes, bit for bit."""
    if part not in (None, "mfma", "f64"):
        raise ValueError("sa_mlp_max: part must be None, 'mfma' or 'f64'")
    feat_kind = int(feat_kind) | (SA_ONLY_MFMA if part == "mfma" else (SA_ONLY_F64 if part == "f64" else 0))
    _chk_f32(xyz, new_xyz, feat, out)
    _chk_i32(idx)
    B, n, _ = xyz.shape
    M, ns = idx.shape[1], idx.shape[2]
    ldf = 0 if feat is None else feat.shape[2]
    args = []
    for (pw, b, g, be) in layers:
        _chk_f32(b, g, be)
        args += [_p(pw.data), _p(b), _p(g), _p(be), pw.cout]
    # 2 FLOP per multiply-add of the three layers over every gathered sample (C + 3 input channels: xyz first)
    flop = 2.0 * B * M * ns * ((C + 3) * layers[0][0].cout + layers[0][0].cout * layers[1][0].cout + layers[1][0].cout * layers[2][0].cout)
    with timed("k:sa_mlp_max%s:%d:%d:%d:%d" % ("" if part is None else "_" + part, C + 3, layers[2][0].cout, B * M * ns, int(flop // 1000000) if part != "f64" else 0), 2):
        # scratch for the register kernel's list of the neighbourhoods the f64 re-evaluation leaves to it (include/caspr_hip.h); the
        # wider shapes take the LDS kernel and no scratch
        ws = None
        if max(l_[0].cout for l_ in layers) <= 64:
            ws = torch.empty(_lib.load().caspr_sa_mlp_max_workspace_ints(B, M), dtype=torch.int32, device=xyz.device)
        _lib.check(_lib.load().caspr_sa_mlp_max_ws_f32(_p(xyz), _p(new_xyz), _p(feat), ldf, _p(idx), B, n, M, C, ns, int(feat_kind), *args,
                                                       _p(out), out.shape[2], out_off, _p(ws), _stream()), "caspr_sa_mlp_max_ws_f32")
    return out


def group_rows_pre(xyz, new_xyz, pre, idx, wx, bias):
    """The pre-aggregated first layer as rows (include/caspr_hip.h: caspr_group_rows_pre_f32): -> Y (B, M*ns, C1) raw layer-1 output."""
    _chk_f32(xyz, new_xyz, pre, wx, bias)
    _chk_i32(idx)
    B, n, _ = xyz.shape
    M, ns = idx.shape[1], idx.shape[2]
    C1 = bias.numel()
    Y = torch.empty(B, M * ns, C1, device=xyz.device, dtype=torch.float32)
    _lib.check(_lib.load().caspr_group_rows_pre_f32(_p(xyz), _p(new_xyz), _p(pre), _chk_rows(pre), _p(idx), B, n, M, C1, ns, _p(wx), _p(bias), _p(Y), C1,
                                                    _stream()), "caspr_group_rows_pre_f32")
    return Y


def sa_mlp_max_pre(xyz, new_xyz, pre, idx, wx, layers, out, out_off):
    """sa_mlp_max with the first layer pre-aggregated (include/caspr_hip.h: caspr_sa_mlp_max_pre_f32): pre (B,n,>=C1) = W_f . feat over the
    level's source points, wx (C1,3) the layer's coordinate columns; layers = 3 x (PackedWeight | None, bias, gamma, beta) -- the first
    entry's weight is not used."""
    _chk_f32(xyz, new_xyz, pre, wx, out)
    _chk_i32(idx)
    B, n, _ = xyz.shape
    M, ns = idx.shape[1], idx.shape[2]
    ldp = _chk_rows(pre)
    (_, b1, g1, be1), (pw2, b2, g2, be2), (pw3, b3, g3, be3) = layers
    _chk_f32(b1, g1, be1, b2, g2, be2, b3, g3, be3)
    C1 = g1.numel()
    # FLOPs of the REFERENCE's formulation (every gathered sample through all three layers), so that the rate stays comparable
    flop = 2.0 * B * M * ns * (C1 * pw2.cout + pw2.cout * pw3.cout)
    with timed("k:sa_mlp_max_pre:%d:%d:%d:%d" % (C1, pw3.cout, B * M * ns, int(flop // 1000000)), 2):
        _lib.check(_lib.load().caspr_sa_mlp_max_pre_f32(_p(xyz), _p(new_xyz), _p(pre), ldp, _p(idx), B, n, M, ns, _p(wx), _p(b1), _p(g1), _p(be1), C1,
                                                        _p(pw2.data), _p(b2), _p(g2), _p(be2), pw2.cout, _p(pw3.data), _p(b3), _p(g3), _p(be3), pw3.cout,
                                                        _p(out), out.shape[2], out_off, _stream()), "caspr_sa_mlp_max_pre_f32")
    return out


LATENT_TEAM = _cfg.latent_team   # multi-workgroup latent ODE kernel (False: single-workgroup kernel)
_team_ws = {}


def _team_workspace(nbytes, device):
    """Dedicated (not shared with other ops) 256-byte aligned scratch of the team kernel: its barrier counters must not
    be recycled by another launch while a solve is in flight on a different stream."""
    key = (device.index, torch.cuda.current_stream().cuda_stream)
    ws = _team_ws.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(nbytes, device=device, dtype=torch.uint8)
        _team_ws[key] = ws
    return ws


# A team barrier that gives up (its workgroups were not co-resident within the spin bound) poisons the solve's output with
# NaN and sets an error word in the workspace.  Reading that word right away would put a host synchronisation back into
# the path, so it is copied to pinned host memory behind the kernel and examined once that copy has completed: at the
# next solve on the same stream, or when the caller asks (check_deferred_errors, e.g. after a synchronize).
_team_status = {}
_LM_WS_STRIDE_WORDS = (256 + (2 * 128 * 16 * 4 + 32 * 4 * 256) * 4) // 4      # LM_WS_STRIDE of csrc/ode.hip, in 32-bit words


_team_pool = []          # pinned status buffers that have been read and may be reused


def _team_raise_if_failed(key, wait=False):
    """Drain every status record of this stream whose copy has completed (all of them with wait=True), oldest first; one record
    per solve, never overwritten before it has been read: a failed solve followed by a successful one still raises."""
    ring = _team_status.get(key)
    failed = False
    while ring:
        host, ev = ring[0]
        if wait:
            ev.synchronize()
        if not ev.query():
            break
        ring.pop(0)
        failed = failed or int(host.max()) != 0
        _team_pool.append(host)
    if failed:
        raise _lib.CasprHipError("caspr_latent_rk4_team_f32: a team barrier gave up (the 32 x ceil(B/16) workgroups were not co-resident "
                                 "within the spin bound); the solve's output was poisoned with NaN.  Set caspr_amd.ops.LATENT_TEAM = False "
                                 "(config.latent_team; it also takes the early solve of reconstruct() off the team kernel) to use the "
                                 "single-workgroup kernel when other streams / processes saturate the GPU")


def check_deferred_errors(wait=True):
    """Raise CasprHipError if an earlier asynchronous kernel reported a failure (the latent team kernel's barrier), or
    CasprAccuracyError / warn if a run-time accuracy check of the fixed-step integrators came back above its tolerance
    (guard_track).  wait=True blocks until the status words of every outstanding solve / check have arrived."""
    for key in list(_team_status):
        _team_raise_if_failed(key, wait=wait)
    _guard_drain(wait=wait)


# ---------------------------------------------------------------------------------------------
# Run-time accuracy guard of the fixed-step integrators (models/caspr.py: CaSPR.check_tol).  The reference's dopri5 bounds its
# integration error at every call (flow.py:96-99, cnf.py:100-119, latent_ode_model.py:38,83); a fixed step count does not.  With the
# guard on, every solve is repeated on a subsample at half (or twice) the step count on a side stream, the two results are compared
# ON THE DEVICE, and the maximum difference travels to pinned host memory behind the comparison -- the same deferred channel as the
# team kernel's error word: no host synchronisation in the path; the verdict is read at the next guarded call or by
# check_deferred_errors().
# ---------------------------------------------------------------------------------------------
class CasprAccuracyError(_lib.CasprHipError):
    """A guarded RK4 solve whose step-halving error estimate exceeds the tolerance asked for (CaSPR.check_tol)."""


_guard_ring = []         # [(pinned host tensor, event, meta)], oldest first
_guard_pool = []
GUARD_LAST = {}          # name -> the last drained record {"diff", "estimate", "tol", "steps", "other_steps", "ok"}
GUARD_HISTORY_MAX = 0.0  # largest estimate / tol ratio seen since reset_guard()


def reset_guard():
    global GUARD_HISTORY_MAX
    GUARD_LAST.clear()
    GUARD_HISTORY_MAX = 0.0


def guard_track(diff, scale, meta):
    """diff: 0-dim float32 device tensor = max |x_S - x_S'| of a check, scale: 0-dim = max |x_S| (both on the CURRENT stream); meta:
    {"name", "tol", "factor", "steps", "other_steps", "action", "what"} -- estimate = factor * diff is compared with
    tol * (1 + scale), i.e. torchdiffeq's  atol + rtol |x|  with atol = rtol = tol (what the reference sets: flow.py:96-99,
    latent_ode_model.py:83), when the copy has arrived."""
    if len(_guard_ring) >= 64:                      # nobody drained for 64 checks: bound the backlog (blocks on the oldest)
        _guard_ring[0][1].synchronize()
    _guard_drain(wait=False)
    host = _guard_pool.pop() if _guard_pool else torch.zeros(2, dtype=torch.float32).pin_memory()
    host.copy_(torch.stack([diff.reshape(()), scale.reshape(())]), non_blocking=True)
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream())
    _guard_ring.append((host, ev, meta))


def _guard_drain(wait=False):
    import math
    import warnings
    global GUARD_HISTORY_MAX
    failed = []
    while _guard_ring:
        host, ev, meta = _guard_ring[0]
        if wait:
            ev.synchronize()
        if not ev.query():
            break
        _guard_ring.pop(0)
        d, xmax = float(host[0]), float(host[1])
        _guard_pool.append(host)
        est = meta["factor"] * d
        bound = meta["tol"] * (1.0 + xmax) if math.isfinite(xmax) else meta["tol"]
        ok = math.isfinite(est) and est <= bound
        GUARD_LAST[meta["name"]] = {"diff": d, "estimate": est, "tol": meta["tol"], "solution_absmax": xmax, "bound": bound, "steps": meta["steps"],
                                    "other_steps": meta["other_steps"], "ok": ok}
        GUARD_HISTORY_MAX = max(GUARD_HISTORY_MAX, est / bound if math.isfinite(est) else float("inf"))
        if not ok:
            msg = ("%s: RK4 with %d steps is not converged to the tolerance asked for: max |x_%d - x_%d| = %.3e on the checked subsample -> "
                   "error estimate %.3e > %.3e = check_tol %.1e x (1 + max |x| %.2f).  Raise the step count (CaSPR.calibrate_rk4_steps picks "
                   "it by step doubling) or the tolerance." % (meta["what"], meta["steps"], meta["steps"], meta["other_steps"], d, est, bound,
                                                              meta["tol"], xmax))
            if meta["action"] == "warn":
                warnings.warn(msg, RuntimeWarning, stacklevel=3)
            else:
                failed.append(msg)
    if failed:
        raise CasprAccuracyError("; ".join(failed))


def _team_track(key, ws, B):
    """Queue the team kernel's error words (one per group of 16 sequences) for a later, non-blocking check (_team_raise_if_failed)."""
    groups = (B + 15) // 16
    words = ws[:groups * _LM_WS_STRIDE_WORDS * 4].view(torch.int32)[16::_LM_WS_STRIDE_WORDS]      # the error word of each group
    ring = _team_status.setdefault(key, [])
    if len(ring) >= 64:                              # nobody drained for 64 solves: bound the backlog (blocks on the oldest)
        ring[0][1].synchronize()
        _team_raise_if_failed(key)
    slot = next((i for i, h in enumerate(_team_pool) if h.numel() == groups), None)
    host = _team_pool.pop(slot) if slot is not None else torch.zeros(groups, dtype=torch.int32).pin_memory()
    ev = torch.cuda.Event()
    host.copy_(words, non_blocking=True)
    ev.record(torch.cuda.current_stream())
    ring.append((host, ev))


def latent_rk4(z0, times, steps, wts, team=None):
    """Fixed-step RK4 of the latent dynamics (latent_ode_model.py:45-70): z0 (B,D) (rows may be a column
    slice of a wider tensor); wts = [PackedWeight0, b0, PackedWeight1, b1, PackedWeight2, b2, PackedWeight3, b3].  -> (B,Tu,D).
    team: True / False forces the 32-workgroup team kernel / the single-workgroup kernel (None: LATENT_TEAM where the shape allows).
    Same values up to the re-association of the layer sums (tests: <= 5e-6)."""
    _chk_f32(times, *[w for w in wts[1::2]])
    if not z0.is_cuda or z0.dtype != torch.float32 or z0.dim() != 2 or z0.stride(1) != 1:
        raise ValueError("latent_rk4: z0 must be a float32 GPU (B,D) tensor with unit column stride")
    B = z0.shape[0]
    D, H = wts[0].cin, wts[0].cout
    if z0.shape[1] != D:
        raise ValueError("latent_rk4: z0 has %d columns, the dynamics net expects %d" % (z0.shape[1], D))
    Tu = times.shape[0]
    out = torch.empty(B, Tu, D, device=z0.device, dtype=torch.float32)
    ptrs = [_p(w.data) if isinstance(w, PackedWeight) else _p(w) for w in wts]
    L = _lib.load()
    if (LATENT_TEAM if team is None else team) and H == 512 and D <= 64 and B <= 64:
        # 32 workgroups per 16 sequences with LDS-resident weights (csrc/ode.hip): the serial chain runs ~3x faster
        key = (z0.device.index, torch.cuda.current_stream().cuda_stream)
        capturing = torch.cuda.is_current_stream_capturing()       # hipGraph capture: no event queries, no host copies
        if not capturing:
            _team_raise_if_failed(key)                   # status of the previous solve on this stream, if it has arrived
        with timed("latent_rk4"):
            ws = _team_workspace(L.caspr_latent_team_ws_bytes(B), z0.device)
            _lib.check(L.caspr_latent_rk4_team_f32(_p(z0), z0.stride(0), _p(times), B, Tu, D, H, int(steps), *ptrs, _p(out), _p(ws), ws.numel(),
                                                   _stream()), "caspr_latent_rk4_team_f32")
        if not capturing:
            _team_track(key, ws, B)
        return out
    with timed("latent_rk4"):
        _lib.check(L.caspr_latent_rk4_f32(_p(z0), z0.stride(0), _p(times), B, Tu, D, H, int(steps), *ptrs, _p(out), _stream()), "caspr_latent_rk4_f32")
    return out



def pack_cnf_x6(w):
    """(512,512) hidden-layer weight of the ODE function -> the three-plane bf16 pack of caspr_cnf_rk4_x6_f32."""
    _chk_f32(w)
    if tuple(w.shape) != (512, 512):
        raise ValueError("pack_cnf_x6: expected a (512,512) weight, got %s" % (tuple(w.shape),))
    out = torch.empty(_lib.load().caspr_cnf_x6_packed_bytes(), device=w.device, dtype=torch.uint8)
    _lib.check(_lib.load().caspr_pack_weight_cnf_x6(_p(w), w.stride(0), _p(out), _stream()), "caspr_pack_weight_cnf_x6")
    return out


CNF_NARROW = 2        # include/caspr_hip.h: CASPR_CNF_NARROW
BEFORE_CNF_LAUNCH = None      # one-shot callable run between a CNF block's hyper-network conv and its launch (models/cnf.py: CNF.integrate);
                              # reconstruct() hands the encoder's deferred T-NOCS regression to it


def cnf_rk4(y, hyper, tcol, w0, b0, w1p, b1, w2p, b2, w3, b3, t_end, steps, reverse, mbn_in=None, mbn_out=None,
            e=None, logp=None, w1x=None, w2x=None, narrow=False):
    """Fixed-step RK4 of one CNF block (cnf.py:70-128).  y (BT,n,3); hyper (BT,ldh).  Returns x or (x, logp).
    w1x / w2x (pack_cnf_x6): when given, the bf16x6 kernel runs the solve (with or without the divergence).
    narrow: the 64-point sampling kernel (a launch that does not fill the chip lasts as long as one workgroup: the accuracy guard)."""
    _chk_f32(y, hyper, tcol, w0, b0, b1, b2, w3, b3, mbn_in, mbn_out, e, logp)
    BT, n, _ = y.shape
    if y.dim() != 3 or y.shape[2] != 3:
        raise ValueError("cnf_rk4: y must be (BT,n,3), got %s" % (tuple(y.shape),))
    if hyper.dim() != 2 or hyper.shape[0] != BT:
        raise ValueError("cnf_rk4: hyper has %s rows for %d frames" % (tuple(hyper.shape), BT))
    if (e is None) != (logp is None):
        raise ValueError("cnf_rk4: the Hutchinson noise e and the initial log-density come together")
    if e is not None and (tuple(e.shape) != tuple(y.shape) or tuple(logp.shape) != (BT, n, 1)):
        raise ValueError("cnf_rk4: e %s / logp %s do not match y %s" % (tuple(e.shape), tuple(logp.shape), tuple(y.shape)))
    for name, m_ in (("mbn_in", mbn_in), ("mbn_out", mbn_out)):
        if m_ is not None and m_.numel() != 12:
            raise ValueError("cnf_rk4: %s must hold 12 floats [weight | bias | running_mean | running_var]" % name)
    out = torch.empty_like(y)
    lp_out = torch.empty(BT, n, 1, device=y.device, dtype=torch.float32) if e is not None else None
    if w1x is not None and w2x is not None:
        with timed("cnf_rk4"):
            _lib.check(_lib.load().caspr_cnf_rk4_x6_f32(_p(y), _p(hyper), hyper.shape[1], _p(tcol), _p(w0), _p(b0), _p(w1x), _p(b1), _p(w2x),
                                                        _p(b2), _p(w3), _p(b3), w0.shape[0], float(t_end), int(steps),
                                                        int(bool(reverse)) | (CNF_NARROW if (narrow and e is None) else 0),
                                                        _p(mbn_in), _p(mbn_out), _p(e), _p(logp), _p(lp_out), _p(out), BT, n, _stream()),
                       "caspr_cnf_rk4_x6_f32")
        return out if e is None else (out, lp_out)
    with timed("cnf_rk4"):
        _lib.check(_lib.load().caspr_cnf_rk4_f32(_p(y), _p(hyper), hyper.shape[1], _p(tcol), _p(w0), _p(b0), _p(w1p.data), _p(b1),
                                                 _p(w2p.data), _p(b2), _p(w3), _p(b3), w0.shape[0], float(t_end), int(steps), int(bool(reverse)),
                                                 _p(mbn_in), _p(mbn_out), _p(e), _p(logp), _p(lp_out), _p(out), BT, n, _stream()),
                   "caspr_cnf_rk4_f32")
    return out if e is None else (out, lp_out)


def chamfer_distance(p, q):
    """tk3dv `ChamferDistance()(pred, gt)` (evaluations.py:40): -> dist1 (B,n), dist2 (B,m) squared NN distances."""
    _chk_f32(p, q)
    B, n, _ = p.shape
    m = q.shape[1]
    d1 = torch.empty(B, n, device=p.device, dtype=torch.float32)
    d2 = torch.empty(B, m, device=p.device, dtype=torch.float32)
    _lib.check(_lib.load().caspr_chamfer_f32(_p(p), _p(q), B, n, m, _p(d1), _p(d2), _stream()), "caspr_chamfer_f32")
    return d1, d2


def earth_mover_distance(xyz1, xyz2, transpose=True):
    """`utils.emd.earth_mover_distance` (emd.py:24-45): approximate EMD cost per cloud pair, (b,3,n) inputs when
    `transpose` (as the reference), else (b,n,3).  -> (b,) ; evaluations.py:46 divides by the point count."""
    if xyz1.dim() == 2:
        xyz1 = xyz1.unsqueeze(0)
    if xyz2.dim() == 2:
        xyz2 = xyz2.unsqueeze(0)
    if transpose:
        xyz1, xyz2 = xyz1.transpose(1, 2), xyz2.transpose(1, 2)
    xyz1, xyz2 = xyz1.contiguous().float(), xyz2.contiguous().float()
    _chk_f32(xyz1, xyz2)
    B, n, _ = xyz1.shape
    m = xyz2.shape[1]
    cost = torch.empty(B, device=xyz1.device, dtype=torch.float32)
    L = _lib.load()
    ws = _workspace(L.caspr_emd_ws_bytes(B, n, m), xyz1.device)
    _lib.check(L.caspr_emd_f32(_p(xyz1), _p(xyz2), B, n, m, _p(cost), _p(ws), ws.numel(), _stream()), "caspr_emd_f32")
    return cost
