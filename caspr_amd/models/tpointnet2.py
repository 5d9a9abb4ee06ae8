"""TPointNet++ encoder (reference: caspr/models/tpointnet2.py) on the HIP kernels.

Same parameter tree (`local_extract`, `global_extract`, `conv1`, `conv2`, `bn1`, `bn2`, `conv3`).
Differences in *how* (not what) it computes, all documented in DESIGN.md:
  * the (B,1088,T*N) tiled global feature (pointnet.py:44-46) is never built: its 1024 constant
    channels become a per-sequence bias of conv1 (W[:,512:1536] . g_b), the 64-channel point
    feature is read in place from the global PointNet's conv1 output;
  * GroupNorm+ReLU between the 1600-wide convs is folded into the next conv's operand load;
  * z0 = max over points of bn2(conv2(.)) comes out of the GroupNorm statistics pass.
"""
import contextlib

import torch
import torch.nn as nn

from .. import ops
from ..config import config as _cfg
from ..utils.weight_cache import WeightCache
from .pointnet import PointNetfeat
from .pointnet2 import PointNet2feat as PointNet2


# True: the deferred T-NOCS regression is queued only when the caller says (reconstruct(): right in front of the flow's launch, through
# ops.BEFORE_CNF_LAUNCH) instead of behind the head's last layer.  Measured and NOT adopted (tools/head_ab.py, profiles/r05_head_ab.txt): queued
# early, this HBM-bound conv holds back the three small kernels the flow waits for (0.7 ms between the encoder's last statistics and the flow's
# first workgroup) -- but queued late it runs entirely in front of the flow: 69.64 -> 69.97 ms.  Early it is.
LATE_TNOCS_LAUNCH = False
GLOBAL_STREAM = _cfg.global_stream   # the global PointNet on a stream of its own beside the index chain and the first set-abstraction kernels
# the head's FIRST layer (576 -> 1600) with its 64-channel remainder beside the main tiles too (ops.conv1x1_gn_tail_beside; no reserved units here:
# the remainder shares them with the persistent kernel).  Measured and NOT adopted (tools/head1_tail_ab.py, outputs identical): 67.30 / 66.82 / 66.41 ms
# with it against 66.71 / 66.60 / 66.39 without -- the remainder's 755 MB of reads slow the tiles down by what it saves behind them.
HEAD1_TAIL_BESIDE = False
# (that remainder on a stream of its OWN from the start of the main tiles, instead of behind the latent solve, was measured too: 71.34 / 71.46 / 71.37 ms
# against 71.51 / 71.44 / 71.52 -- within the noise; not kept)
TAIL_BESIDE = True     # the last head layer's 64-channel remainder on the early solve's stream / compute units (ops.conv1x1_gn_early)


def _tensors(obj):
    if torch.is_tensor(obj):
        yield obj
    elif isinstance(obj, dict):
        for v in obj.values():
            yield from _tensors(v)
    elif isinstance(obj, (list, tuple)):
        for v in obj:
            yield from _tensors(v)


class TPointNet2(nn.Module):
    def __init__(self, radii_list=[0.02, 0.05, 0.1, 0.2, 0.4, 0.8], local_feat_size=512, out_feat_size=1600,
                 augment_quad=True, augment_pairs=True, tnocs_point_size=4, regress_tnocs=True):
        super(TPointNet2, self).__init__()
        self.augment_quad = augment_quad
        self.augment_pairs = augment_pairs
        self.tnocs_point_size = tnocs_point_size
        self.local_feat_size = local_feat_size
        self.local_bottleneck_size = self.local_feat_size
        self.global_feat_size = 1024
        self.space_time_pt_feat = 64
        self.latent_feat_size = out_feat_size

        in_features = 0
        if self.augment_quad:
            in_features += 3
        if self.augment_pairs:
            in_features += 3
        self.local_extract = PointNet2(in_features=in_features, num_classes=self.local_feat_size, batchnorm=False,
                                       use_xyz_feature=True, use_random_ball_query=False, radii_list=radii_list,
                                       max_feat_prop_size=self.local_bottleneck_size)
        self.global_extract = PointNetfeat(input_dim=4, out_size=self.global_feat_size)

        per_point_out_size = self.global_feat_size + self.space_time_pt_feat + self.local_feat_size
        self.conv1 = torch.nn.Conv1d(per_point_out_size, per_point_out_size, 1)
        self.conv2 = torch.nn.Conv1d(per_point_out_size, self.latent_feat_size, 1)
        self.bn1 = nn.GroupNorm(16, per_point_out_size)
        self.bn2 = nn.GroupNorm(16, self.latent_feat_size)
        self.regress_tnocs = regress_tnocs
        if self.regress_tnocs:
            self.conv3 = torch.nn.Conv1d(self.latent_feat_size, self.tnocs_point_size, 1)
            self.loss_func = torch.nn.L1Loss(reduction='none')   # reference: L1Loss(reduce=False), tpointnet2.py:68
        self._cache = WeightCache()
        self.record = None  # set to a list to capture FPS / ball-query indices (parity tests)

    def _side_stream(self, device, which=0):
        # (a CU-masked side stream -- hipExtStreamCreateWithCUMask leaving one unit in eight to the latent team -- was tried in
        # round 3: the mask had no measurable effect on where kernels ran, and the step got 3 ms slower with the external stream)
        key = (device.type, device.index, which)
        if not hasattr(self, "_streams"):
            self._streams = {}
        if key not in self._streams:
            self._streams[key] = torch.cuda.Stream(device=device)
        return self._streams[key]

    def _head_weights(self):
        L, G, S = self.local_feat_size, self.global_feat_size, self.space_time_pt_feat

        def build():
            w = self.conv1.weight.detach()[:, :, 0]
            w_pt = torch.cat([w[:, :L], w[:, L + G:]], dim=1).contiguous()       # columns of [local | point feature]
            w_g = w[:, L:L + G].contiguous()                                     # columns of the tiled global feature
            # PointNet++'s last layer is a plain conv (pointnet2.py:247: no norm, no activation behind it) and conv1 is linear in
            # it: conv1(W_f h + b_f) = (W1_local W_f) h + W1_local b_f.  Folded weight (f64 product, rounded once) + bias term: the
            # 512 -> 512 layer over all B T N points (0.9 ms of the cfg-2 step) is never run
            f3 = self.local_extract.final_layers[3]
            wl = w[:, :L].double()
            w_fold = torch.cat([(wl @ f3.weight.detach()[:, :, 0].double()).float(), w[:, L + G:]], dim=1).contiguous()
            b_fold = (wl @ f3.bias.detach().double()).float().contiguous()
            return ops.PackedWeight(w_pt), ops.PackedWeight(w_g), ops.PackedWeight(w_fold), b_fold
        f3_ = self.local_extract.final_layers[3]
        p1 = self._cache.get("conv1", [self.conv1.weight, f3_.weight, f3_.bias], build)
        p2 = self._cache.get("conv2", [self.conv2.weight],
                             lambda: ops.PackedWeight(self.conv2.weight.detach()[:, :, 0].contiguous()))
        p3 = None
        if self.regress_tnocs:
            p3 = self._cache.get("conv3", [self.conv3.weight],
                                 lambda: ops.PackedWeight(self.conv3.weight.detach()[:, :, 0].contiguous()))
        return p1, p2, p3

    def launch_tnocs(self):
        """Queue the T-NOCS regression that forward(x, defer_tnocs=True) prepared (:105-106) on the side stream, behind everything the
        current stream holds NOW.  reconstruct() calls it through ops.BEFORE_CNF_LAUNCH, i.e. between the flow's hyper-network conv and the
        flow's launch: queued earlier, this HBM-bound conv (2,560 workgroups on every compute unit for 0.5 ms) held back the three small
        kernels the flow waits for -- a 1 MB cat took 166 us, the hyper conv 105 us, 0.2 ms of nothing in front of them."""
        pend = getattr(self, "_tnocs_pending", None)
        if pend is None:
            return
        self._tnocs_pending = None
        p3, y2, s2, t2, out, side = pend
        main = torch.cuda.current_stream()
        side.wait_stream(main)
        with torch.cuda.stream(side):
            ops.conv1x1(p3, self.conv3.bias, y2, in_scale=s2, in_shift=t2, in_relu=True, act=1, out=out)  # :105-106
            self._tnocs_ready = torch.cuda.Event()
            self._tnocs_ready.record()
        for t_ in (y2, s2, t2, out):
            t_.record_stream(side)          # main-stream allocations the side stream still reads / writes

    def join(self):
        """Make the current stream wait for a T-NOCS regression that forward(x, defer_tnocs=True) left to the side stream (queueing it
        now if nobody has).  No-op when nothing is pending."""
        self.launch_tnocs()
        ev = getattr(self, "_tnocs_ready", None)
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)
            self._tnocs_ready = None

    def forward(self, x, defer_tnocs=False, early=None):
        """x (B,T,N,4) -> z0 (B, out_feat_size), tnocs (B,T,N,4) | None   (tpointnet2.py:70-115).
        defer_tnocs: the T-NOCS regression (:105-106; nothing downstream of z0 needs it) is issued on the side stream and
        the caller continues with z0 on the current one; the caller must call join() before anything reads tnocs.
        early: callable(z0_partial) invoked as soon as the first `early.channels` columns of z0 are final -- after the first
        channel tile of the last head layer (ops.conv1x1_gn_early) -- so that the caller can start the latent solve beside
        the rest of that layer; not called when the layer does not run in pieces (the caller then proceeds as usual)."""
        if not x.is_cuda:
            raise ValueError("caspr_amd.TPointNet2 runs on the GPU only (HIP kernels); got a %s tensor" % x.device)
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            # autograd is recording (train() or eval() alike, as in the reference): one autograd node with a taped forward and a HIP backward (train/encoder_grad.py)
            from ..train.encoder_grad import encode_with_grad
            return encode_with_grad(self, x)
        B, T, N, _ = x.size()
        x = x.contiguous().float()
        L, S = self.local_feat_size, self.space_time_pt_feat
        P = T * N
        # one buffer holds the head's input [local (L) | raw global conv1 output (S)]
        X1 = torch.empty(B, P, L + S, device=x.device, dtype=torch.float32)
        # local branch, part 1: every index tensor (FPS / ball query / three-NN) depends on xyz only.  The chain is
        # latency-bound (one 256-thread workgroup per frame, 1,872 dependent rounds), so it runs on a side stream
        # underneath the MFMA-bound global PointNet below.
        C = (3 if self.augment_quad else 0) + (3 if self.augment_pairs else 0)
        main = torch.cuda.current_stream()
        xyz, feat = ops.prep_input(x, self.augment_quad, self.augment_pairs)
        side = self._side_stream(x.device)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            idx = self.local_extract.indices(xyz, events=True)
        if C == 0:
            feat = None
        # global spatio-temporal feature (tpointnet2.py:75-76).  GLOBAL_STREAM: on a stream of its own, joined in front of the head -- the
        # set abstraction then starts when the first level's indices are ready (0.9 ms with the LDS ball query) instead of behind these
        # convs (1.2 ms), which run beside its latency-bound first kernels
        gstream = None
        if GLOBAL_STREAM and self.record is None and not torch.cuda.is_current_stream_capturing():
            gstream = self._side_stream(x.device, 1)
            gstream.wait_stream(main)
        with (torch.cuda.stream(gstream) if gstream is not None else contextlib.nullcontext()):
            with ops.timed("enc_global_pointnet"):
                pf, gmax = self.global_extract.features(x.view(B, P, 4), y1_out=X1[:, :, L:])
        if gstream is not None:
            X1.record_stream(gstream)
            x.record_stream(gstream)
        # no join here: local_extract.run waits for each level's indices where it uses them
        for t_ in _tensors(idx):
            t_.record_stream(main)
        # local spatial feature per time step (tpointnet2.py:79-93)
        with ops.timed("enc_local_pointnet2"):
            kind = (ops.FEAT_QUAD if self.augment_quad else 0) | (ops.FEAT_PAIRS if self.augment_pairs else 0)
            # fold: PointNet++'s last (purely linear) layer lives inside conv1's weight (_head_weights), so the local branch stops at
            # that layer's operand -- the raw output of final_layers[0] with its per-FRAME GroupNorm + ReLU still to be applied
            fold = self.record is None
            loc = self.local_extract.run(xyz, feat, C, out=X1.view(B * T, N, L + S)[:, :, :L], record=self.record, idx=idx, feat_kind=kind,
                                         stop_before_last=fold)
        if gstream is not None:
            main.wait_stream(gstream)
            for t_ in (pf.scale, pf.shift, gmax):
                if t_ is not None:
                    t_.record_stream(main)
        t_head = ops.timed("enc_head")
        t_head.__enter__()

        (w_pt, w_g, w_fold, b_fold), p2, p3 = self._head_weights()
        # conv1 over [local | global max (tiled) | point feature]  (tpointnet2.py:96-99)
        bbias = ops.conv1x1(w_g, self.conv1.bias, gmax.view(B, 1, -1))                        # (B,1,1600)
        if fold:
            # the conv runs over FRAMES (its local operand is normalised per frame), its own statistics pool the T frames of a
            # sequence (bn1 normalises over all T N points); the per-sequence rows are repeated per frame (small (B T, .) tensors)
            _, s_f, t_f = loc
            rep = lambda v: v.repeat_interleave(T, dim=0)
            in_scale = torch.cat([s_f, rep(pf.scale)], dim=1).contiguous()
            in_shift = torch.cat([t_f, rep(pf.shift)], dim=1).contiguous()
            bb = rep(bbias.view(B, -1)[:, :self.conv1.out_channels] + b_fold).contiguous()
            # (its 64-channel remainder beside the main tiles, as the next layer's: ops.conv1x1_gn_tail_beside)
            y1, s1, t1 = ops.conv1x1_gn_tail_beside(w_fold, None, X1.view(B * T, N, L + S), self.bn1.weight, self.bn1.bias,
                                                    self._side_stream(x.device, 2) if HEAD1_TAIL_BESIDE else None, bbias=bb, in_scale=in_scale,
                                                    in_shift=in_shift, in_relu=True, in_relu_from=0, pool=T)
            y1 = y1.view(B, P, y1.shape[2])
        else:
            ones = torch.ones(B, L, device=x.device, dtype=torch.float32)
            in_scale = torch.cat([ones, pf.scale], dim=1).contiguous()
            in_shift = torch.cat([torch.zeros_like(ones), pf.shift], dim=1).contiguous()
            y1, s1, t1 = ops.conv1x1_gn(w_pt, None, X1, self.bn1.weight, self.bn1.bias, bbias=bbias.view(B, -1), in_scale=in_scale,
                                        in_shift=in_shift, in_relu=True, in_relu_from=L)
        if early is not None and self.record is None and ops.conv1x1_gn_early_ok(p2, B, P, 16, early.channels):
            y2, s2, t2, z0 = ops.conv1x1_gn_early(p2, self.conv2.bias, y1, self.bn2.weight, self.bn2.bias, early, in_scale=s1, in_shift=t1,
                                                  in_relu=True, reserve_cus=early.reserve_cus(B),
                                                  tail_stream=early.stream if TAIL_BESIDE else None)   # :99-100, 111
        else:
            y2, s2, t2, z0 = ops.conv1x1_gn(p2, self.conv2.bias, y1, self.bn2.weight, self.bn2.bias, want_max=True,
                                            in_scale=s1, in_shift=t1, in_relu=True)               # :99-100, 111
        del y1
        tnocs_regression = None
        if self.regress_tnocs and defer_tnocs and self.record is None:
            # prepared here, queued by launch_tnocs() (the caller's choice of moment; join() at the latest)
            t = torch.empty(B, P, (p3.cout + 3) // 4 * 4, device=x.device, dtype=torch.float32)
            self._tnocs_pending = (p3, y2, s2, t2, t, side)
            if not LATE_TNOCS_LAUNCH:
                self.launch_tnocs()
            tnocs_regression = t[:, :, :self.tnocs_point_size].reshape(B, T, N, self.tnocs_point_size)
        elif self.regress_tnocs:
            t = ops.conv1x1(p3, self.conv3.bias, y2, in_scale=s2, in_shift=t2, in_relu=True, act=1)  # :105-106
            tnocs_regression = t[:, :, :self.tnocs_point_size].reshape(B, T, N, self.tnocs_point_size)
        t_head.__exit__()
        return z0, tnocs_regression

    def loss(self, outputs, gt):
        """Per-element L1 (tpointnet2.py:117-122)."""
        return self.loss_func(outputs, gt)
