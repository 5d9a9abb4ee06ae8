"""PointNet++ MSG feature extractor (reference: caspr/models/pointnet2.py) on the HIP kernels.

Same module / parameter tree as the reference (`set_abstractions.{l}.pointnet_modules.{s}.conv_layers`,
`.bn_layers`, `feature_propagators.{l}.unit_pointnet.{0,1,3,4}`, `final_layers.{0,1,3}`), so reference
checkpoints load unchanged.  The Kaolin operators (pointnet2.py:7) are replaced by `caspr_amd.ops`;
grouping + per-neighbourhood MLP is one fused kernel; GroupNorm+ReLU between pointwise convs is
folded into the consumer's operand load (`Lazy`).
Internally everything is point-major (B, P, C); the sub-module `forward`s keep the reference's
channels-first signatures by transposing at the boundary.
"""
import contextlib
from typing import Iterable

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from ..config import config as _cfg
from ..utils.weight_cache import WeightCache
from .lazy import Lazy

NUM_GROUPS = 16  # for group norm (pointnet2.py:12)
LO_PARTS = _cfg.sa_lo_parts  # the first set-abstraction level hands its output to the second as an unevaluated sum hi + lo (PointNet2feat.run)
LONG_SCALE_ON_MAIN = True              # of a level's two scales the one with MORE samples on the caller's stream, the other on the side stream (False: rounds 5's order)
BALL_QUERY_PAIR = True                 # the two ball queries of a level as one launch (ops.ball_query_pair)
SCALE_STREAMS = _cfg.sa_scale_streams  # the two scales of a set-abstraction level on two streams (PointNet2SetAbstraction.run)
PRE_AGGREGATE = _cfg.sa_pre_aggregate  # the wide set-abstraction levels' first layer once per source point (PointNet2SetAbstraction.run)
FP_COMMUTE = _cfg.fp_commute           # feature propagation's first conv on the coarse level where the skip part is tiny (PointNet2FeaturePropagator.run)
F64_STREAMS = _cfg.sa_f64_streams      # ... and the f64 re-evaluation of each scale's small balls on a stream of its own, beside its MFMA kernel
_SCALE_STREAM = {}


def _scale_stream(device, which=0):
    key = (device.type, device.index, which)
    if key not in _SCALE_STREAM:
        _SCALE_STREAM[key] = torch.cuda.Stream(device=device)
    return _SCALE_STREAM[key]


def separate_xyz_and_features(points):
    """Kaolin helper (pointnet2.py:228): (B,n,3+C) -> xyz (B,n,3), features (B,C,n) | None."""
    xyz = points[..., 0:3].contiguous()
    features = points[..., 3:].transpose(1, 2).contiguous() if points.size(-1) > 3 else None
    return xyz, features


class PointNet2GroupingLayer(nn.Module):
    """Parameter-free grouper (ball query + gather + centre subtraction + xyz||feat), pointnet2.py:340-342."""

    def __init__(self, radius, num_samples, use_xyz_feature=True, use_random_ball_query=False):
        super(PointNet2GroupingLayer, self).__init__()
        if use_random_ball_query:
            raise ValueError("use_random_ball_query=True is not supported (the reference runs with False, tpointnet2.py:49)")
        if not use_xyz_feature:
            raise ValueError("use_xyz_feature=False is not supported (the reference runs with True, tpointnet2.py:48)")
        self.radius = radius
        self.num_samples = num_samples

    def forward(self, xyz, new_xyz, features=None):
        """xyz (B,n,3), new_xyz (B,M,3), features (B,C,n) -> (B, M, 3+C, ns) (reference layout)."""
        idx = ops.ball_query(self.radius, self.num_samples, xyz.contiguous(), new_xyz.contiguous())
        feat_pm = None if features is None else features.transpose(1, 2).contiguous()
        return ops.group_points(xyz.contiguous(), new_xyz.contiguous(), feat_pm, idx)


class PointNetFeatureExtractor(nn.Module):
    """Per-neighbourhood PointNet (pointnet2.py:527-703); only the configuration the reference
    instantiates (global_feat=True, transposed_input=True, GroupNorm) runs on the fused kernel."""

    def __init__(self, in_channels: int = 3, feat_size: int = 1024, layer_dims: Iterable[int] = [64, 128],
                 global_feat: bool = True, activation=F.relu, batchnorm: bool = True, transposed_input: bool = False):
        super(PointNetFeatureExtractor, self).__init__()
        if not isinstance(in_channels, int):
            raise TypeError('Argument in_channels expected to be of type int. Got {0} instead.'.format(type(in_channels)))
        if not isinstance(feat_size, int):
            raise TypeError('Argument feat_size expected to be of type int. Got {0} instead.'.format(type(feat_size)))
        if not hasattr(layer_dims, '__iter__'):
            raise TypeError('Argument layer_dims is not iterable.')
        for idx, layer_dim in enumerate(layer_dims):
            if not isinstance(layer_dim, int):
                raise TypeError('Elements of layer_dims must be of type int. Found type {0} at index {1}.'.format(type(layer_dim), idx))
        if not isinstance(global_feat, bool):
            raise TypeError('Argument global_feat expected to be of type bool. Got {0} instead.'.format(type(global_feat)))
        if batchnorm:
            raise ValueError("batchnorm=True is not supported: the reference uses GroupNorm (tpointnet2.py:47)")
        self.feat_size = feat_size
        self.global_feat = global_feat
        layer_dims = list(layer_dims)
        layer_dims.insert(0, in_channels)
        layer_dims.append(feat_size)
        self.conv_layers = nn.ModuleList()
        self.bn_layers = nn.ModuleList()
        for idx in range(len(layer_dims) - 1):
            self.conv_layers.append(nn.Conv1d(layer_dims[idx], layer_dims[idx + 1], 1))
            self.bn_layers.append(nn.GroupNorm(NUM_GROUPS, layer_dims[idx + 1]))
        self.batchnorm = batchnorm
        self.transposed_input = transposed_input
        self.in_channels = in_channels
        self._cache = WeightCache()

    def kernel_layers(self):
        """3 x (PackedWeight, bias, gamma, beta).  The first layer's K is permuted to the gather order
        of caspr_sa_mlp_max_f32: [features (C, zero-padded to a multiple of 4) | dx dy dz]."""
        if len(self.conv_layers) != 3:
            raise ValueError("the fused set-abstraction kernel runs 3-layer point MLPs (pointnet2.py:62-146)")
        out = []
        for l, (conv, gn) in enumerate(zip(self.conv_layers, self.bn_layers)):
            def build(conv=conv, l=l):
                w = conv.weight.detach()[:, :, 0]
                if l == 0:
                    C = w.shape[1] - 3
                    pad = (-C) % 4
                    w = torch.cat([w[:, 3:], w.new_zeros(w.shape[0], pad), w[:, :3]], dim=1)
                return ops.PackedWeight(w.contiguous())
            out.append((self._cache.get("l%d" % l, [conv.weight], build), conv.bias, gn.weight, gn.bias))
        return out

    def pre_layers(self):
        """The first layer split for pre-aggregation (csrc/sa_mlp.hip): (PackedWeight of its feature columns W[:, 3:], its coordinate
        columns W[:, :3] as a (C1, 3) tensor)."""
        conv = self.conv_layers[0]

        def build():
            w = conv.weight.detach()[:, :, 0]
            return ops.PackedWeight(w[:, 3:].contiguous()), w[:, :3].contiguous()
        return self._cache.get("pre", [conv.weight], build)

    def row_layers(self, cin_pad):
        """3 x (PackedWeight, bias, gamma, beta) for the row-materialised form (run_rows): reference channel order
        [dx dy dz | features], the first layer's input width zero-padded to cin_pad."""
        out = []
        for l, (conv, gn) in enumerate(zip(self.conv_layers, self.bn_layers)):
            def build(conv=conv, l=l):
                w = conv.weight.detach()[:, :, 0]
                if l == 0 and cin_pad > w.shape[1]:
                    w = torch.nn.functional.pad(w, (0, cin_pad - w.shape[1]))
                return ops.PackedWeight(w.contiguous())
            out.append((self._cache.get(("rows", l, cin_pad), [conv.weight], build), conv.bias, gn.weight, gn.bias))
        return out

    def forward(self, x):
        """Reference signature (B', C, ns) -> (B', feat_size): one neighbourhood per row."""
        if not (self.global_feat and self.transposed_input):
            raise ValueError("only global_feat=True, transposed_input=True is supported (pointnet2.py:352-359)")
        Bp, C, ns = x.shape
        # neighbourhood tensor -> fake cloud: every sample is a point, centre at the origin
        pts = x.transpose(1, 2).contiguous()                      # (B', ns, C)
        xyz = pts[:, :, :3].contiguous()
        Cf = C - 3
        ldf = (Cf + 3) // 4 * 4
        feat = None
        if Cf > 0:
            feat = pts.new_zeros(Bp, ns, ldf)
            feat[:, :, :Cf] = pts[:, :, 3:]
        idx = torch.arange(ns, device=x.device, dtype=torch.int32).view(1, 1, ns).repeat(Bp, 1, 1).contiguous()
        centre = torch.zeros(Bp, 1, 3, device=x.device, dtype=torch.float32)
        out = torch.empty(Bp, 1, self.feat_size, device=x.device, dtype=torch.float32)
        ops.sa_mlp_max(xyz, centre, feat, idx, Cf, self.kernel_layers(), out, 0)
        return out.view(Bp, self.feat_size)


# Set-abstraction scales whose point MLP reads at least this many channels run row-materialised (_run_rows) instead of fused: at
# cfg-2 that is the coarsest level (515 inputs, 16 centres per frame: 13.97 vs 14.31 ms for the PointNet++ stage); one level
# finer (259 inputs) the extra passes over the rows cost more than the bf16 pipe returns (14.16 ms).
ROWS_MIN_CIN = 500


class PointNet2SetAbstraction(nn.Module):
    """Set-abstraction level with multi-scale grouping (pointnet2.py:253-422)."""

    def __init__(self, num_points_out, pointnet_in_features, pointnet_layer_dims_list, radii_list=None,
                 num_samples_list=None, batchnorm=True, use_xyz_feature=True, use_random_ball_query=False):
        super(PointNet2SetAbstraction, self).__init__()
        if num_points_out is None:
            raise ValueError("num_points_out=None (group-all) is not used by the reference model and not supported")
        assert isinstance(radii_list, list) and isinstance(num_samples_list, list), 'radii_list and num_samples_list must be lists'
        assert (len(radii_list) == len(num_samples_list) == len(pointnet_layer_dims_list)), (
            'Dimension of radii_list ({}), num_samples_list ({}), pointnet_layer_dims_list ({}) must match'
            .format(len(radii_list), len(num_samples_list), len(pointnet_layer_dims_list)))
        self.num_points_out = num_points_out
        self.pointnet_layer_dims_list = pointnet_layer_dims_list
        self.grouper_modules = nn.ModuleList()
        self.pointnet_modules = nn.ModuleList()
        self.layers = []
        self.pointnet_in_channels = pointnet_in_features + (3 if use_xyz_feature else 0)
        for i in range(len(radii_list)):
            pointnet_layer_dims = pointnet_layer_dims_list[i]
            assert isinstance(pointnet_layer_dims, list), 'Each pointnet_layer_dims must be a list, got {} instead'.format(pointnet_layer_dims)
            assert len(pointnet_layer_dims) > 0, 'Each pointnet_layer_dims must have at least one element'
            self.grouper_modules.append(PointNet2GroupingLayer(radii_list[i], num_samples_list[i], use_xyz_feature=use_xyz_feature,
                                                               use_random_ball_query=use_random_ball_query))
            self.pointnet_modules.append(PointNetFeatureExtractor(in_channels=self.pointnet_in_channels, feat_size=pointnet_layer_dims[-1],
                                                                  layer_dims=pointnet_layer_dims[:-1], global_feat=True,
                                                                  batchnorm=batchnorm, transposed_input=True))
            self.layers.append(num_samples_list[i])

    def get_num_features_out(self):
        return sum([lst[-1] for lst in self.pointnet_layer_dims_list])

    def indices(self, xyz, events=False):
        """FPS centres + both ball queries of this level (depend on xyz only): -> dict(fps_idx, new_xyz, ball_idx).
        events=True (called on a side stream): "scale_ready"[i] is recorded behind ball query i, so that scale i's kernel can start
        while the next query still runs (at the first level the consumer is waiting: 0.24 ms of the step)."""
        fps_idx, new_xyz = ops.furthest_point_sampling(xyz, self.num_points_out, return_xyz=True)   # pointnet2.py:384-387
        # the query of the scale with the MOST samples first: its point MLP is the long one of the level (twice the columns), and with
        # the scales on two streams (run()) the level ends when that one does -- it should not wait behind the other scale's query
        ball, ready = [None] * len(self.layers), [None] * len(self.layers)
        order = sorted(range(len(self.layers)), key=lambda i_: -self.layers[i_])
        if BALL_QUERY_PAIR and len(self.layers) == 2:
            # both scales' queries in one pass over the cloud (round 6: at the first level they were 0.50 + 0.38 ms in front of the first
            # set-abstraction kernel; one distance per (centre, point) serves both radii) -- the same rows, bit for bit
            g0, g1 = self.grouper_modules
            ball[0], ball[1] = ops.ball_query_pair(g0.radius, self.layers[0], g1.radius, self.layers[1], xyz, new_xyz)      # :391
            if events:
                ready[0] = ready[1] = torch.cuda.Event()
                ready[0].record()
        else:
            for i in order:
                ball[i] = ops.ball_query(self.grouper_modules[i].radius, self.layers[i], xyz, new_xyz)  # :391
                if events:
                    ready[i] = torch.cuda.Event()
                    ready[i].record()
        d = {"fps_idx": fps_idx, "new_xyz": new_xyz, "ball_idx": ball}
        if events:
            d["scale_ready"] = ready
            d["last_scale"] = order[-1]
            d["first_scale"] = order[0]
        return d

    @staticmethod
    def _await(idx, scale=None):
        """When the indices were computed on another stream (PointNet2feat.indices(events=True)), wait for THIS level's -- for
        one scale's ball query if the level carries per-scale events."""
        evs = idx.get("scale_ready")
        ev = evs[scale] if (evs and scale is not None) else idx.get("ready")
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)

    def narrow(self):
        """All point MLPs of this level run on the register kernel (every width <= 64): the shapes that can exchange low parts."""
        return all(max(d) <= 64 for d in self.pointnet_layer_dims_list)

    def run(self, xyz, feat, C, record=None, idx=None, feat_kind=0, lo_in=False, lo_out=False):
        """Point-major core: xyz (B,n,3), feat (B,n,ldf) with C valid channels (or None).
        -> new_xyz (B,M,3), new_feat (B,M,Cout).  `idx` = precomputed self.indices(xyz); feat_kind: see ops.sa_mlp_max.
        lo_out: new_feat is (B,M,2 Cout) = [features | their low parts] (ops.FEAT_LO_OUT: what the f32 value lacks of the kernel's f64
        result); lo_in: `feat` is such a tensor of the previous level, C its feature count (ops.FEAT_LO_IN).  narrow() levels only."""
        B = xyz.shape[0]
        M = self.num_points_out
        if idx is None:
            idx = self.indices(xyz)
        if "scale_ready" not in idx:
            self._await(idx)
        new_xyz = idx["new_xyz"]
        if (lo_in or lo_out) and not self.narrow():
            raise ValueError("low parts are exchanged by the register set-abstraction kernel only (all widths <= 64)")
        feat_kind = feat_kind | (ops.FEAT_LO_IN if lo_in else 0) | (ops.FEAT_LO_OUT if lo_out else 0)
        out = torch.empty(B, M, self.get_num_features_out() * (2 if lo_out else 1), device=xyz.device, dtype=torch.float32)
        # The scales of a level are independent (they read the same input and write disjoint column ranges of `out`): the LAST one
        # is issued on a second stream, so that one scale's latency-bound phases (GroupNorm / max chains, the f64 re-evaluation of
        # small balls on the vector pipe) run beside the other's MFMAs; joined before the level returns.
        main = torch.cuda.current_stream()
        side = None
        if SCALE_STREAMS and xyz.is_cuda and len(self.layers) > 1 and not torch.cuda.is_current_stream_capturing():
            side = _scale_stream(xyz.device)
            side.wait_stream(main)
        # On the register kernel's levels a scale is two independent halves since the kernel works from a list (csrc/sa_mlp.hip:
        # sa_list_kernel): the MFMA kernel over the neighbourhoods the f64 re-evaluation does not take, and that re-evaluation (three
        # latency-bound launches on the vector pipe).  F64_STREAMS puts the f64 half on a stream of its own per scale, beside the MFMA
        # half.  OFF by default: measured (tools/sa_streams_ab.py, profiles/r05_sa_f64_streams_ab.txt) the five levels take 7.13-7.28 ms
        # that way against 6.74-6.84 with the re-evaluation behind its kernel -- four resident kernels time-slice the wave slots the two
        # scales already fill, and every extra stream costs its joins.
        halves = side is not None and self.narrow() and F64_STREAMS
        off, extra = 0, []
        # which scale goes to the side stream: the one with FEWER samples (round 6).  The side stream shares a hardware queue with other side
        # streams of the step (HIP multiplexes a process's streams onto four queues; more queues measured slower: profiles/r06_hwq_ab.txt) --
        # at the first level with the global PointNet's convs, behind which a kernel queued there waits (profiles/r06c_step_timeline.txt:
        # the 32-sample scale started 1.3 ms after its indices were ready) -- so the LONG kernel of the level runs on the caller's stream
        side_scale = (min(range(len(self.layers)), key=lambda i_: (self.layers[i_], -i_)) if LONG_SCALE_ON_MAIN else len(self.layers) - 1) if side is not None else None
        for i, ns in enumerate(self.layers):
            on_side = side is not None and i == side_scale
            with (torch.cuda.stream(side) if on_side else contextlib.nullcontext()):
                if "scale_ready" in idx:
                    self._await(idx, i)
                if ops.CONV_BF16X6 and C + 3 >= ROWS_MIN_CIN and (M * ns) % 128 == 0:
                    self._run_rows(xyz, new_xyz, feat, C, idx["ball_idx"][i], i, out, off)
                elif PRE_AGGREGATE and self.pointnet_layer_dims_list[i][0] >= 64 and C >= 32 and feat_kind == 0 and xyz.shape[1] >= 128:
                    # the first layer's feature part once per SOURCE point (a conv over this level's n points) instead of once per
                    # (centre, sample) pair; the fused kernel starts from it (csrc/sa_mlp.hip: pre-aggregated first layer)
                    pn = self.pointnet_modules[i]
                    pw_f, wx = pn.pre_layers()
                    pre = ops.conv1x1(pw_f, None, feat)
                    ops.sa_mlp_max_pre(xyz, new_xyz, pre, idx["ball_idx"][i], wx, pn.kernel_layers(), out, off)
                else:
                    ops.sa_mlp_max(xyz, new_xyz, feat, idx["ball_idx"][i], C, self.pointnet_modules[i].kernel_layers(), out, off,
                                   feat_kind=feat_kind, part="mfma" if halves else None)  # :391-409
            if halves:
                st = _scale_stream(xyz.device, 1 + i)
                st.wait_stream(main)
                with torch.cuda.stream(st):
                    if "scale_ready" in idx:
                        self._await(idx, i)
                    ops.sa_mlp_max(xyz, new_xyz, feat, idx["ball_idx"][i], C, self.pointnet_modules[i].kernel_layers(), out, off,
                                   feat_kind=feat_kind, part="f64")
                extra.append(st)
            off += self.pointnet_layer_dims_list[i][-1]
        if side is not None:
            main.wait_stream(side)
        for st in extra:
            main.wait_stream(st)
            for t_ in (xyz, new_xyz, feat, out) + tuple(idx["ball_idx"]):
                if t_ is not None:
                    t_.record_stream(st)
        if record is not None:
            record.append(idx)
        return new_xyz, out

    def _run_rows(self, xyz, new_xyz, feat, C, ball_idx, i, out, off):
        """One scale with materialised neighbourhood rows: gather -> 3 x (bf16x6 conv -> per-neighbourhood GroupNorm [-> ReLU]) ->
        max over the samples.  For the widest levels (few centres, inputs of 500+ channels) the fused kernel streams the whole
        point MLP (1.3 MB of weights at the coarsest level) from L2 once per neighbourhood; as pointwise convs over all
        neighbourhood rows the weights are shared by 256-row tiles and the products run on the bf16 pipe."""
        from .. import train_ops as T
        ns = self.layers[i]
        pn = self.pointnet_modules[i]
        kin = (3 + C + 31) // 32 * 32
        layers = pn.row_layers(kin)
        pre_agg = PRE_AGGREGATE and C % 4 == 0 and self.pointnet_layer_dims_list[i][0] % 4 == 0
        if pre_agg:
            # the first layer's feature part once per source point (64 rows per frame at the coarsest level instead of 256 / 512
            # gathered ones), then its rows = pre[sample] + W_x (p - c) + b straight from the gather (csrc/backward_points.hip)
            pw_f, wx = pn.pre_layers()
            pre = ops.conv1x1(pw_f, None, feat)
            cur = None
        else:
            cur = T.group_rows(xyz, new_xyz, feat, C, ball_idx, align=32)
        for l, (pw, bias, gamma, beta) in enumerate(layers):
            y = ops.group_rows_pre(xyz, new_xyz, pre, ball_idx, wx, bias) if (pre_agg and l == 0) else ops.conv1x1(pw, bias, cur)
            last = l == len(layers) - 1
            cur, _, _, _ = T.gn_rows(y, ns, pw.cout, gamma, beta, relu=not last, maxout=out[:, :, off:off + pw.cout] if last else None)

    def forward(self, xyz, features=None):
        """Reference signature: xyz (B,n,3), features (B,C,n) -> new_xyz (B,M,3), new_features (B,Cout,M)."""
        feat, C = None, 0
        if features is not None:
            C = features.shape[1]
            feat = features.new_zeros(features.shape[0], features.shape[2], (C + 3) // 4 * 4)
            feat[:, :, :C] = features.transpose(1, 2)
        new_xyz, out = self.run(xyz.contiguous(), feat, C)
        return new_xyz, out.transpose(1, 2)


class PointNet2FeaturePropagator(nn.Module):
    """Feature propagation level (pointnet2.py:424-528)."""

    def __init__(self, num_features, num_features_prev, layer_dims, batchnorm=True):
        super(PointNet2FeaturePropagator, self).__init__()
        if batchnorm:
            raise ValueError("batchnorm=True is not supported: the reference uses GroupNorm (tpointnet2.py:47)")
        self.layer_dims = layer_dims
        unit_pointnets = []
        in_features = num_features + num_features_prev
        for out_features in layer_dims:
            unit_pointnets.append(nn.Conv1d(in_features, out_features, 1))
            unit_pointnets.append(nn.GroupNorm(NUM_GROUPS, out_features))
            unit_pointnets.append(nn.ReLU())
            in_features = out_features
        self.unit_pointnet = nn.Sequential(*unit_pointnets)
        self._cache = WeightCache()

    def get_num_features_out(self):
        return self.layer_dims[-1]

    def _packed(self, i, cin_pad=None):
        """cin_pad: the conv's input width padded with zero columns (a multiple of 32, for the bf16x6 kernel)."""
        conv = self.unit_pointnet[i]

        def build():
            w = conv.weight.detach()[:, :, 0]
            if cin_pad is not None and cin_pad > w.shape[1]:
                w = torch.nn.functional.pad(w, (0, cin_pad - w.shape[1]))
            return ops.PackedWeight(w.contiguous())
        return self._cache.get((i, cin_pad), [conv.weight], build)

    def run(self, xyz, xyz_prev, feat, C, prev: Lazy, nn=None):
        """Point-major core.  feat (B,n,ldf) skip features with C valid channels (or None); prev = Lazy
        features of the coarser level; nn = precomputed ops.three_nn(xyz, xyz_prev, with_weights=True).  -> Lazy (B,n,Cout)."""
        _, idx, w = nn if nn is not None else ops.three_nn(xyz, xyz_prev, with_weights=True)    # pointnet2.py:514-518
        n_layers = len(self.layer_dims)
        conv0, gn0 = self.unit_pointnet[0], self.unit_pointnet[1]
        # The first conv on the COARSE level (interpolation and a pointwise conv commute, csrc/gemm.hip: three_interp_add_gn_kernel):
        # worth it where the fine level has at least twice the rows and the skip part is a handful of channels (the finest level's six
        # augmented coordinates: a 544 -> 512 conv over 327,680 rows becomes a 512 -> 512 conv over 163,840 at cfg-2).  A rule on the
        # level's shape only -- never on the number of frames -- so sharding stays bitwise invariant.
        if (FP_COMMUTE and ops.CONV_BF16X6 and C <= 8 and 2 * prev.raw.shape[1] <= idx.shape[1] and prev.raw.shape[1] % 128 == 0
                and prev.channels % 32 == 0 and prev.channels >= 192 and conv0.out_channels % 64 == 0 and 256 % (conv0.out_channels // 4) == 0):
            Cp = prev.channels

            def build():
                W = conv0.weight.detach()[:, :, 0]
                return ops.PackedWeight(W[:, :Cp].contiguous()), W[:, Cp:Cp + C].contiguous()
            pw_p, w_s = self._cache.get(("commute", Cp, C), [conv0.weight], build)
            u = ops.conv1x1(pw_p, None, prev.raw, in_scale=prev.scale, in_shift=prev.shift, in_relu=prev.relu)
            y, s, t = ops.three_interp_add_gn(u, idx, w, feat, C, w_s if C else None, conv0.bias, gn0.weight, gn0.bias)
            cur = Lazy(y, conv0.out_channels, s, t, True)
            for l in range(1, n_layers):                                                        # :525
                conv, gn = self.unit_pointnet[3 * l], self.unit_pointnet[3 * l + 1]
                y, s, t = ops.conv1x1_gn(self._packed(3 * l), conv.bias, cur.raw, gn.weight, gn.bias, in_scale=cur.scale, in_shift=cur.shift,
                                         in_relu=cur.relu)
                cur = Lazy(y, conv.out_channels, s, t, True)
            return cur
        # an input width the bf16x6 conv cannot take (518 at the finest level) is padded with zero columns to a multiple of 32
        cin = prev.channels + C
        pad = ops.CONV_BF16X6 and cin % 32 != 0 and cin >= 192 and idx.shape[1] % 128 == 0
        x = ops.three_interpolate(prev.raw, idx, w, skip=feat, skip_channels=C, in_scale=prev.scale,
                                  in_shift=prev.shift, in_relu=prev.relu, C=prev.channels, align=32 if pad else 4)   # :519-523
        cur = Lazy(x, cin)
        n_layers = len(self.layer_dims)
        for l in range(n_layers):                                                               # :525
            conv, gn = self.unit_pointnet[3 * l], self.unit_pointnet[3 * l + 1]
            pw = self._packed(3 * l, x.shape[2]) if (pad and l == 0) else self._packed(3 * l)
            y, s, t = ops.conv1x1_gn(pw, conv.bias, cur.raw, gn.weight, gn.bias, in_scale=cur.scale, in_shift=cur.shift,
                                     in_relu=cur.relu)
            cur = Lazy(y, conv.out_channels, s, t, True)
        return cur

    def forward(self, xyz, xyz_prev, features=None, features_prev=None):
        """Reference signature (channels-first features) -> (B, Cout, n)."""
        if xyz_prev is None:
            raise ValueError("xyz_prev=None (global feature broadcast) is not used by the reference model and not supported")
        feat, C = None, 0
        if features is not None:
            C = features.shape[1]
            feat = features.new_zeros(features.shape[0], features.shape[2], (C + 3) // 4 * 4)
            feat[:, :, :C] = features.transpose(1, 2)
        fp = features_prev.transpose(1, 2).contiguous()
        out = self.run(xyz.contiguous(), xyz_prev.contiguous(), feat, C, Lazy(fp, fp.shape[2]))
        return out.materialize().transpose(1, 2)


class PointNet2feat(nn.Module):
    """Modified PointNet++ segmentation network giving per-point features (pointnet2.py:14-249)."""

    def __init__(self, in_features=0, num_classes=2, batchnorm=True, use_xyz_feature=True, use_random_ball_query=False,
                 radii_list=[0.02, 0.05, 0.1, 0.2, 0.4, 0.8], max_feat_prop_size=512):
        super(PointNet2feat, self).__init__()
        if len(radii_list) != 6:
            raise ValueError('Radii list must be length 6, not %d!' % (len(radii_list)))
        self.in_features = in_features
        # (num_points_out, [mlp scale A, mlp scale B]) -- pointnet2.py:62-146 with batchnorm=False
        specs = [
            (1024, [[16, 16, 32], [32, 32, 64]]),
            (512, [[32, 32, 64], [32, 32, 64]]),
            (256, [[64, 64, 128], [64, 96, 128]]),
            (64, [[128, 196, 256] if batchnorm else [128, 256, 256], [128, 196, 256] if batchnorm else [128, 256, 256]]),
            (16, [[256, 256, 512], [256, 384, 512] if batchnorm else [256, 256, 512]]),
        ]
        self.set_abstractions = nn.ModuleList()
        feats = in_features
        for l, (m, dims) in enumerate(specs):
            sa = PointNet2SetAbstraction(num_points_out=m, pointnet_in_features=feats, pointnet_layer_dims_list=dims,
                                         radii_list=[radii_list[l], radii_list[l + 1]], num_samples_list=[16, 32],
                                         batchnorm=batchnorm, use_xyz_feature=use_xyz_feature,
                                         use_random_ball_query=use_random_ball_query)
            self.set_abstractions.append(sa)
            feats = sa.get_num_features_out()

        self.feature_propagators = nn.ModuleList()
        fp_div = [1, 1, 2, 2, 4]                                                                # pointnet2.py:150-193
        prev = self.set_abstractions[-1].get_num_features_out()
        for l in range(5):
            layer_dims = [max([max_feat_prop_size // fp_div[l], num_classes])] * 2
            nf = self.set_abstractions[-2 - l].get_num_features_out() if l < 4 else in_features
            fp = PointNet2FeaturePropagator(num_features=nf, num_features_prev=prev, layer_dims=layer_dims, batchnorm=batchnorm)
            self.feature_propagators.append(fp)
            prev = fp.get_num_features_out()

        final_dim = layer_dims[0]
        self.final_layers = nn.Sequential(
            nn.Conv1d(self.feature_propagators[-1].get_num_features_out(), final_dim, 1),
            nn.GroupNorm(NUM_GROUPS, final_dim),
            nn.ReLU(),
            nn.Conv1d(final_dim, num_classes, 1))
        self.num_classes = num_classes
        self._cache = WeightCache()

    def _packed_final(self, i):
        conv = self.final_layers[i]
        return self._cache.get(i, [conv.weight], lambda: ops.PackedWeight(conv.weight.detach()[:, :, 0].contiguous()))

    def indices(self, xyz, events=False):
        """Every index tensor of the level stack -- FPS, ball queries, three-NN + weights -- depends on the
        coordinates only.  Computing them up front lets the caller run this latency-bound chain (1,872 dependent
        FPS rounds per frame, one small workgroup per frame) on a side stream under the MFMA-bound kernels.
        events=True (called on that side stream): every level's dict carries a "ready" event, so the consumer starts a level's
        set abstraction as soon as ITS indices exist while the chain continues underneath (run() waits per level)."""
        sa_idx, xyz_list = [], [xyz]
        for sa in self.set_abstractions:
            d = sa.indices(xyz_list[-1], events=events)
            if events:
                d["ready"] = d["scale_ready"][d["last_scale"]]          # the whole level: its last ball query
            sa_idx.append(d)
            xyz_list.append(d["new_xyz"])
        nn = []
        target = -2
        for _ in self.feature_propagators:
            nn.append(ops.three_nn(xyz_list[target], xyz_list[target + 1], with_weights=True))
            target -= 1
        out = {"sa": sa_idx, "nn": nn}
        if events:
            out["nn_ready"] = torch.cuda.Event()
            out["nn_ready"].record()
        return out

    def run(self, xyz, feat, C, out=None, record=None, idx=None, feat_kind=0, stop_before_last=False):
        """Point-major core: xyz (B,n,3), feat (B,n,ldf) with C valid channels.  -> (B,n,num_classes)
        written into `out` (may be a column slice of a wider buffer) if given.  `idx` = precomputed self.indices(xyz);
        feat_kind: what the input features of the FIRST level are (ops.FEAT_QUAD | ops.FEAT_PAIRS), see ops.sa_mlp_max.
        stop_before_last: return (raw output of final_layers[0] [in `out`], its GroupNorm scale, shift) -- the operand of the last,
        purely linear layer (pointnet2.py:247), for a caller that folds that layer into its own first layer (TPointNet2)."""
        if idx is None:
            idx = self.indices(xyz)
        xyz_list, feat_list, ch_list = [xyz], [feat], [C]
        # the first level hands its output to the second as hi + lo (both run on the register kernel, whose f64 reference column and f64
        # re-evaluation of small balls would otherwise start from f32-rounded inputs: DESIGN.md section 5, round 5); everything else
        # reads the hi half through a strided view
        sas = self.set_abstractions
        lo_from = [l + 1 < len(sas) and sas[l].narrow() and sas[l + 1].narrow() and LO_PARTS and l == 0 for l in range(len(sas))]
        for l, sa in enumerate(sas):                                                            # pointnet2.py:232
            lo_in = l > 0 and lo_from[l - 1]
            if "scale_ready" in idx["sa"][l]:
                # (the stage clock below starts when the level's FIRST index rows exist -- what the level's first kernel waits for anyway --
                # not when the host reaches this line: with the global PointNet on its own stream the main stream arrives here a
                # millisecond before the first level's ball query is done, and that wait belongs to the index chain)
                sa._await(idx["sa"][l], idx["sa"][l].get("first_scale"))
            with ops.timed("enc_set_abstraction"):       # wall time of the level on the main stream (its scales overlap on two streams)
                xyz, feat = sa.run(xyz, feat, C, record, idx["sa"][l], feat_kind=feat_kind if l == 0 else 0, lo_in=lo_in, lo_out=lo_from[l])
            C = sa.get_num_features_out()
            xyz_list.append(xyz)
            feat_list.append(feat[:, :, :C] if lo_from[l] else feat)                           # consumers other than the next level: hi only
            ch_list.append(C)
        prev = Lazy(feat_list[-1], ch_list[-1])
        target = -2
        if idx.get("nn_ready") is not None:
            torch.cuda.current_stream().wait_event(idx["nn_ready"])
        for l, fp in enumerate(self.feature_propagators):                                       # :238-245
            prev = fp.run(xyz_list[target], xyz_list[target + 1], feat_list[target], ch_list[target], prev, idx["nn"][l])
            target -= 1
        c0, gn, c3 = self.final_layers[0], self.final_layers[1], self.final_layers[3]
        y, s, t = ops.conv1x1_gn(self._packed_final(0), c0.bias, prev.raw, gn.weight, gn.bias, in_scale=prev.scale, in_shift=prev.shift,
                                 in_relu=prev.relu, out=out if stop_before_last else None)
        if stop_before_last:
            return y, s, t
        return ops.conv1x1(self._packed_final(3), c3.bias, y, in_scale=s, in_shift=t, in_relu=True, out=out)  # :247

    def forward(self, points):
        """Reference signature: points (B,n,3+in_features) -> (B,n,num_classes)."""
        xyz = points[..., 0:3].contiguous()
        C = points.shape[-1] - 3
        feat = None
        if C > 0:
            feat = points.new_zeros(points.shape[0], points.shape[1], (C + 3) // 4 * 4)
            feat[:, :, :C] = points[..., 3:]
        out = self.run(xyz, feat, C)
        return out[:, :, :self.num_classes].contiguous()
