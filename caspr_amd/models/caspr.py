"""CaSPR model surface (reference: caspr/models/caspr.py) on the MI355X HIP kernels.

Drop-in for `caspr.models.caspr.CaSPR`: same constructor arguments, methods, return tuples and
state_dict keys, so train.py / test.py / viz.py-style callers and reference
checkpoints work unchanged.  Extra keyword-only knobs of this build: `cnf_rk4_steps`,
`latent_rk4_steps` (fixed-step RK4 replaces torchdiffeq's adaptive dopri5).

Inference: `encode`, `reconstruct`, `decode`, and `forward` for NLL / T-NOCS loss values.
Training: in train() mode with grad enabled `forward` is differentiable end to end -- the encoder as one autograd node
with a HIP backward (caspr_amd/train/encoder_grad.py), the latent ODE and the CNF through the discrete RK4 map with
every matrix product on the HIP kernels (caspr_amd/train/flow_grad.py).
"""
import weakref

import numpy as np
import torch
import torch.nn as nn

from .. import ops
from ..config import config as _cfg
from .tpointnet2 import TPointNet2
from .latent_ode_model import LatentODE
from .flow import get_point_cnf, count_nfe, PointCNFArgs
from .utils import standard_normal_logprob, sample_gaussian, sphere_surface_points, truncated_normal


class _EarlyLatent:
    """The latent solve started from inside the encoder's last layer (TPointNet2.forward(early=...)): the ODE's initial state is the
    first `channels` columns of z0 (caspr.py:169), final after that layer's first channel tile; the solve then runs on a side stream
    BESIDE the layer's remaining two thirds -- on the team kernel with 32 RESERVED compute units per 16 sequences (its LDS-resident
    weights make it immune to the layer's L2 traffic: 2.65 ms), or, when the team kernel is switched off (ops.LATENT_TEAM = False or
    config.early_latent_team = False), on the single-workgroup kernel with one reserved unit -- instead of 2.3 ms in front of the flow
    with the rest of the chip idle.  The kernel is the one every other solve of the process uses (ops.LATENT_TEAM decides both), so a
    sequence's latent codes do not depend on which path solved them."""

    def __init__(self, latent_ode, plan, stream):
        self.latent_ode, self.plan, self.stream = latent_ode, plan, stream
        self.channels = latent_ode.input_size
        self.out, self.event = None, None

    @staticmethod
    def team():
        return bool(ops.LATENT_TEAM and EARLY_LATENT_TEAM)

    def reserve_cus(self, B):
        return (32 if self.team() else 1) * ((B + 15) // 16)

    def __call__(self, z0_partial):
        main = torch.cuda.current_stream()
        self.stream.wait_stream(main)
        with torch.cuda.stream(self.stream):
            z_init = z0_partial[:, :self.channels]
            out = ops.latent_rk4(z_init, self.plan["sorted_t"], self.latent_ode.rk4_steps, self.latent_ode._weights(), team=self.team())
            # what aggregate_and_solve_latent does with the solution, here too: off the path between the encoder's last kernel and the flow
            self.out = out[self.plan["rows"], self.plan["pos"], :]                      # (B, T, H): the requested stamps
            self.latent_ode.ode_func._num_evals.mul_(0).add_(self.plan["evals"])        # evaluations actually run (an element-wise op, not a blit)
            self.event = torch.cuda.Event()
            self.event.record()
        z0_partial.record_stream(self.stream)
        self.out.record_stream(main)


# the latent solve beside the encoder's last layer (config.early_latent = False: in front of the flow, as in rounds 1-3)
JOIN_TNOCS_LATE = True        # reconstruct(): join the deferred T-NOCS regression behind the flow's launch (False: in front of it, rounds 2-4)
EARLY_LATENT = _cfg.early_latent
EARLY_LATENT_TEAM = _cfg.early_latent_team
_EARLY_STREAM = {}
_EARLY_DRAW = {}      # (id(model), device, sample size) -> pinned buffer, copy stream, last copy's event (CaSPR._draw_early)


def _drop_early_draw(owner):
    for k in [k for k in _EARLY_DRAW if k[0] == owner]:
        _EARLY_DRAW.pop(k, None)


_GUARD_STREAM = {}
_UNSET = object()


def _guard_stream(device):
    key = str(device)
    if key not in _GUARD_STREAM:
        _GUARD_STREAM[key] = torch.cuda.Stream(device=device)
    return _GUARD_STREAM[key]


def _other_steps(S):
    """The step count a guarded solve is compared with, and the factor that turns max |x_S - x_other| into the error estimate of the
    S-step result (RK4, global error ~ C h^4): S' = S // 2 steps when S >= 2 -- x_S' - x_S = ((S / S')^4 - 1) e_S, so estimate =
    diff / ((S / S')^4 - 1): 1 / 15 for an even S, and the same cheap half-cost check for an odd one (S = 11 -> S' = 5, 1 / 22.4) --
    else twice the steps: x_1 - x_2 = (15 / 16) e_1, estimate = diff * 16 / 15."""
    if S >= 2:
        S2 = S // 2
        return S2, 1.0 / ((float(S) / S2) ** 4 - 1.0)
    return 2 * S, 16.0 / 15.0


def _first_passing(diff_of, candidates, tol, refine=True, max_tries=3):
    """Step-count search of calibrate_rk4_steps.  diff_of(S) = max |x_S - x_2S| (15/16 of the S-step error of a 4th-order method).
    Walk `candidates` in ascending order to the first count that passes (diff <= tol); then, with `refine`, look BETWEEN the last
    failing candidate and it: predict the smallest passing count from the two measured differences (observed order p = log2 of their
    ratio per doubling, clipped to [2, 4]; diff(S) ~ diff(fail) (fail / S)^p), and VERIFY the prediction by an actual S-vs-2S solve,
    moving up one count at a time (at most `max_tries` solves) -- a trained flow that fails at 8 and passes at 16 usually passes at
    11 or 12, which is a quarter fewer function evaluations than 16.  -> (chosen, {S: diff} of every count tried)."""
    import math
    diffs, chosen, fail = {}, None, None
    for S in sorted(candidates):
        diffs[S] = diff_of(S)
        if diffs[S] <= tol:
            chosen = S
            break
        fail = S
    if chosen is None:
        return max(candidates), diffs
    if refine and fail is not None and chosen - fail > 1:
        d_fail, d_pass = diffs[fail], diffs[chosen]
        order = 4.0
        if math.isfinite(d_fail) and d_pass > 0 and d_fail > d_pass:
            order = min(4.0, max(2.0, math.log(d_fail / d_pass) / math.log(float(chosen) / fail)))
        guess = chosen
        if math.isfinite(d_fail) and d_fail > 0:
            guess = max(fail + 1, int(math.ceil(fail * (d_fail / tol) ** (1.0 / order) - 1e-9)))
        S, tries = guess, 0
        while S < chosen and tries < max_tries:
            diffs[S] = diff_of(S)
            tries += 1
            if diffs[S] <= tol:
                chosen = S
                break
            S += 1
    return chosen, diffs


class CaSPR(nn.Module):
    def __init__(self, radii_list=[0.02, 0.05, 0.1, 0.2, 0.4, 0.8], local_feat_size=512, latent_feat_size=1600,
                 ode_hidden_size=512, motion_feat_size=64, pretrain_tnocs=False, augment_quad=True, augment_pairs=True,
                 cnf_blocks=1, regress_tnocs=True, *, cnf_rk4_steps=8, latent_rk4_steps=2, check_tol=1e-5, latent_check_tol=None,
                 check_action="warn", check_points=64):
        super(CaSPR, self).__init__()
        # Run-time accuracy guard of the fixed-step integrators: ON by default at the reference's own tolerances, reporting as a
        # RuntimeWarning (check_action="raise": CasprAccuracyError; check_tol=None: off).  The reference's dopri5 controls its error at every
        # call (CNF atol = rtol = 1e-5, flow.py:96-99; latent ODE 1e-3, latent_ode_model.py:38,83); a fixed step count does not -- a trained
        # checkpoint loaded at the default 8 / 2 steps must not be silently under-resolved (calibrate_rk4_steps installs counts that pass).
        # With check_tol set, every inference solve (reconstruct / decode / forward / aggregate_and_solve_latent under no_grad) is repeated at half
        # the step count on `check_points` samples per frame (the latent solve: all of it, on one compute unit) on a side stream, the
        # results are compared on the device, and the Richardson estimate of the delivered solution's error is examined through
        # ops.check_deferred_errors() / at the next guarded call: CasprAccuracyError (check_action "raise") or a RuntimeWarning
        # ("warn") when it exceeds check_tol x (1 + max |x|) -- torchdiffeq's atol + rtol |x| with atol = rtol = check_tol, as the
        # reference sets them (latent: latent_check_tol, default 100 x check_tol = the reference's ratio).
        self.check_tol = check_tol
        self.latent_check_tol = latent_check_tol
        self.check_action = check_action
        self.check_points = check_points
        self.pretrain_tnocs = pretrain_tnocs
        self.augment_quad = augment_quad
        self.augment_pairs = augment_pairs
        self.motion_feat_size = motion_feat_size
        self.regress_tnocs = regress_tnocs
        self.tnocs_point_size = 4
        self.encoder = TPointNet2(radii_list, local_feat_size=local_feat_size, out_feat_size=latent_feat_size,
                                  augment_quad=self.augment_quad, augment_pairs=self.augment_pairs,
                                  tnocs_point_size=self.tnocs_point_size, regress_tnocs=self.regress_tnocs)
        if self.pretrain_tnocs:
            return
        self.latent_ode = LatentODE(input_size=self.motion_feat_size, hidden_size=ode_hidden_size, num_layers=2,
                                    nonlinearity=nn.Tanh, rk4_steps=latent_rk4_steps)
        self.cnf_args = PointCNFArgs()
        self.cnf_args.zdim = latent_feat_size
        self.cnf_args.num_blocks = cnf_blocks
        self.cnf_args.rk4_steps = cnf_rk4_steps
        self.point_cnf = get_point_cnf(self.cnf_args)

    # ------------------------------------------------------------------------------------------
    def forward(self, x, sample_points, aggregate_points=None, e=None):
        """caspr.py:76-122.  x, sample_points (B,T,N,4) -> (recon_loss (B,T,N), tnocs_loss (B,T,N,4)).
        `e` (B*T,N,3) optionally fixes the Hutchinson noise (odefunc.py:115-117)."""
        if self._differentiable(x, sample_points):
            return self._forward_train(x, sample_points, e)
        with torch.no_grad():
            z0, tnocs_pred = self.encode(x)
            B, H = z0.size()
            _, T, N, _ = sample_points.size()
            tnocs_loss = None
            if self.regress_tnocs:
                tnocs_loss = self.encoder.loss(tnocs_pred[:, :, :, :self.tnocs_point_size],
                                               sample_points[:, :, :, :self.tnocs_point_size])
            if self.pretrain_tnocs:
                return tuple([tnocs_loss])
            ode_feat_dim = self.cnf_args.zdim
            all_times = sample_points[:, :, 0, 3]                                               # :106
            sample_feats = self.aggregate_and_solve_latent(z0, all_times)
            z = sample_feats.reshape(B * T, ode_feat_dim)
            pts = sample_points.reshape(B * T, N, 4)[:, :, :3].contiguous()                     # :112
            init_logprob = torch.zeros(B * T, N, 1, device=pts.device, dtype=pts.dtype)
            guard = None
            if self.check_tol is not None and pts.is_cuda and not self.training:
                if e is None:
                    e = torch.randn_like(pts)                                                   # what the solve would draw (odefunc.py:127-128)
                guard = self._guard_cnf_begin(pts, z, init_logprob, e.reshape(B * T, N, -1))
            cnf_result = self.point_cnf(pts, z, init_logprob, e=e)
            if guard is not None:
                self._guard_cnf_end(guard, cnf_result)
            recon_loss = self.get_nll_loss(cnf_result, B, T)
            return tuple([recon_loss, tnocs_loss])

    def _differentiable(self, *inputs):
        """The reference's forward is differentiable whenever autograd records (train() or eval() alike; its callers wrap
        evaluation in torch.no_grad(), train.py:152, test.py:124,141).  Same rule here: the taped path runs iff grad mode
        is on and a parameter or an input asks for a gradient; under no_grad the inference kernels run."""
        if not torch.is_grad_enabled():
            return False
        return any(t.requires_grad for t in inputs if torch.is_tensor(t)) or any(p.requires_grad for p in self.parameters())

    def _forward_train(self, x, sample_points, e=None):
        """Differentiable forward (train_utils.py:125 calls it under model.train()): the encoder is one autograd node with
        a HIP backward (train/encoder_grad.py); the latent ODE and the CNF differentiate the RK4 map (train/flow_grad.py)."""
        from ..train.flow_grad import point_cnf_train
        z0, tnocs_pred = self.encode(x)
        B, _ = z0.size()
        _, T, N, _ = sample_points.size()
        tnocs_loss = None
        if self.regress_tnocs:
            tnocs_loss = self.encoder.loss(tnocs_pred[:, :, :, :self.tnocs_point_size], sample_points[:, :, :, :self.tnocs_point_size])
        if self.pretrain_tnocs:
            return tuple([tnocs_loss])
        all_times = sample_points[:, :, 0, 3]
        sample_feats = self.aggregate_and_solve_latent(z0, all_times)
        z = sample_feats.reshape(B * T, self.cnf_args.zdim)
        pts = sample_points.reshape(B * T, N, 4)[:, :, :3].contiguous()
        init_logprob = torch.zeros(B * T, N, 1, device=pts.device, dtype=pts.dtype)
        cnf_result = point_cnf_train(self.point_cnf, pts, z, init_logprob, e=e)
        return tuple([self.get_nll_loss(cnf_result, B, T), tnocs_loss])

    def get_nll_loss(self, cnf_result_list, B, T):
        """caspr.py:124-146."""
        batch_size = B * T
        y, delta_log_py = cnf_result_list
        cloud_dim = y.size()[1]
        log_py = standard_normal_logprob(y).sum(2)
        delta_log_py = delta_log_py.view(batch_size, cloud_dim)
        log_px = log_py - delta_log_py
        return (-log_px).view((B, T, -1))

    def encode(self, x):
        """caspr.py:148-155."""
        return self.encoder(x)

    def aggregate_and_solve_latent(self, z0, time_tensor, _plan=None, _presolved=None):
        """caspr.py:157-183: unique sorted times -> latent ODE -> map back -> concat the static feature.
        Inference on the GPU takes the synchronisation-free route of LatentODE.solve_at (same values); _plan: what
        reconstruct() prepared ahead of the encoder (LatentODE.plan_times)."""
        B, T = time_tensor.size()
        z_init = z0[:, :self.latent_ode.input_size]
        z_global = z0[:, self.latent_ode.input_size:]
        if _presolved is not None and _presolved.event is not None:
            # the solve already ran beside the encoder's last layer (_EarlyLatent): join it, gather the requested stamps
            torch.cuda.current_stream().wait_event(_presolved.event)
            sample_feats = _presolved.out
        elif z0.is_cuda and not self._differentiable(z0, time_tensor):
            if _plan is not None and _plan.get("event") is not None:
                torch.cuda.current_stream().wait_event(_plan["event"])                   # the plan was made on another stream
                for v in _plan.values():
                    if torch.is_tensor(v):
                        v.record_stream(torch.cuda.current_stream())
            sample_feats = self.latent_ode.solve_at(z_init, time_tensor, _plan)
        else:
            solve_t, time_map = torch.unique(time_tensor, sorted=True, return_inverse=True)
            pred_z = self.gen_latent(z_init, solve_t)
            batch_inds = torch.arange(B, device=z0.device).view((-1, 1)).repeat((1, T))
            sample_feats = pred_z[batch_inds, time_map, :]
        if self.check_tol is not None and z0.is_cuda and not self._differentiable(z0, time_tensor):
            self._guard_latent(z_init, time_tensor, sample_feats, _plan)
        B_global, H_global = z_global.size()
        z_global = z_global.unsqueeze(1).expand(B_global, sample_feats.size()[1], H_global)
        return torch.cat([sample_feats, z_global], dim=2)

    # ------------------------------------------------------------------------------------------ run-time accuracy guard
    def _guard_latent(self, z_init, time_tensor, sample_feats, plan):
        """The latent solve once more at half (or twice) the steps per interval, on the single-workgroup kernel (one compute unit,
        no co-residency requirement: it may run beside anything) on the guard stream; max |z_L - z_L'| over the requested stamps
        goes to the deferred channel (ops.guard_track).  Nothing on the current stream waits for it."""
        L = self.latent_ode.rk4_steps
        L2, factor = _other_steps(L)
        tol = self.latent_check_tol if self.latent_check_tol is not None else 100.0 * self.check_tol
        main = torch.cuda.current_stream()
        gs = _guard_stream(z_init.device)
        gs.wait_stream(main)
        with torch.cuda.stream(gs), ops.untimed():
            if plan is None or plan["shape"] != tuple(time_tensor.shape):
                plan = self.latent_ode.plan_times(time_tensor)
            zc = ops.latent_rk4(z_init, plan["sorted_t"], L2, self.latent_ode._weights(), team=False)[plan["rows"], plan["pos"], :]
            diff = (sample_feats - zc).abs().amax()
            ops.guard_track(diff, sample_feats.abs().amax(), {"name": "latent", "tol": float(tol), "factor": factor, "steps": L, "other_steps": L2, "action": self.check_action,
                                   "what": "latent ODE (latent_ode_model.py:45-70; reference: dopri5 at rtol = atol = 1e-3)"})
        # everything the guard stream reads was allocated on another stream (the plan: on the plan / copy stream): keep it alive until
        # the check has run, also when reconstruct() drops the plan right after this call
        for t_ in (z_init, sample_feats, time_tensor) + tuple(v for v in plan.values() if torch.is_tensor(v)):
            t_.record_stream(gs)

    def _guard_cnf_begin(self, y, z, logpx=None, e=None):
        """What the check of this solve needs, captured BEFORE the main launch (the weight packs / end time are built on first use: on the
        MAIN stream, before the guard stream reads them).  -> what _guard_cnf_end needs.
        logpx / e given: the density direction of forward() (cnf.py:70-128 with the Hutchinson divergence, same noise on the same
        samples); otherwise the sampling direction of decode()."""
        from .cnf import CNF
        blocks = [l for l in self.point_cnf.chain if isinstance(l, CNF)]
        for b in blocks:
            b._weights()
            if ops.CNF_BF16X6:
                b._weights_x6()
            b.end_time()
        S = blocks[0].rk4_steps
        S2, factor = _other_steps(S)
        return {"y": y, "z": z, "logpx": logpx, "e": e, "blocks": blocks, "g": min(int(self.check_points), y.shape[1]), "S": S, "S2": S2, "factor": factor,
                "stream": _guard_stream(y.device)}

    def _guard_cnf_end(self, ctx, x):
        """The point CNF once more on the first `check_points` samples of every frame at half (S = 1: twice) the step count on the guard
        stream, queued BEHIND the main launch (round 6; rounds 5 queued it beside the launch, where each of its 160 workgroups held up one
        of the 256 compute units' ten main workgroups: +3 % of a step): the check then runs under whatever the caller does next -- in a loop
        over batches the first, latency-bound milliseconds of the next call's encoder (farthest-point sampling on 160 of 256 units) -- and
        its verdict, which travels through the deferred channel anyway (ops.guard_track), arrives a millisecond later.
        max |x_S - x_S'| on the checked samples -> the deferred channel; nothing on the current stream waits."""
        gs, g, blocks = ctx["stream"], ctx["g"], ctx["blocks"]
        y, z, logpx, e = ctx["y"], ctx["z"], ctx["logpx"], ctx["e"]
        done = torch.cuda.Event()
        done.record(torch.cuda.current_stream())
        saved = [(b, b.rk4_steps) for b in blocks]
        with torch.cuda.stream(gs), ops.untimed():
            gs.wait_event(done)
            try:
                for b in blocks:
                    b.rk4_steps, b._count_evals, b.odefunc._count_evals, b._narrow = _other_steps(b.rk4_steps)[0], False, False, g <= 64
                if logpx is None:
                    xh = self.point_cnf(y[:, :g].contiguous(), z, reverse=True)
                else:
                    xh = self.point_cnf(y[:, :g].contiguous(), z, logpx[:, :g].contiguous(), e=e[:, :g].contiguous())
            finally:
                for b, st in saved:
                    b.rk4_steps, b._count_evals, b.odefunc._count_evals, b._narrow = st, True, True, False
            meta = {"tol": float(self.check_tol), "factor": ctx["factor"], "steps": ctx["S"], "other_steps": ctx["S2"], "action": self.check_action}
            if torch.is_tensor(x):
                diff = (x[:, :g] - xh).abs().amax()
                ops.guard_track(diff, x[:, :g].abs().amax(), dict(meta, name="cnf", what="point CNF (cnf.py:70-128; reference: dopri5 at atol = rtol = 1e-5)"))
            else:               # density direction: the state is (y, logp), both integrated (cnf.py:112-126)
                for i, (nm, what) in enumerate((("cnf_fwd_y", "point CNF, density direction, y"), ("cnf_fwd_logp", "point CNF, density direction, log-density"))):
                    diff = (x[i][:, :g] - xh[i]).abs().amax()
                    ops.guard_track(diff, x[i][:, :g].abs().amax(), dict(meta, name=nm, what=what + " (cnf.py:70-128; reference: dopri5 at atol = rtol = 1e-5)"))
        for t_ in (y, z, logpx, e) + ((x,) if torch.is_tensor(x) else tuple(x)):
            if t_ is not None:
                t_.record_stream(gs)

    def gen_latent(self, z0, timestamps):
        """caspr.py:185-196."""
        return self.latent_ode(z0, timestamps)

    def get_nfe(self):
        """caspr.py:198-202."""
        return np.array([count_nfe(self.latent_ode), count_nfe(self.point_cnf)])

    def _draw_early(self, B, T, num_points, constant_in_time, device):
        """The Gaussian base samples of decode, drawn as the reference draws them -- torch.randn on the CPU generator, then moved
        (models/utils.py:25-26; caspr.py:252) -- but EARLY in reconstruct(): the encoder's kernels are already queued, so the
        ~7 ms draw of a cfg-2 batch (983,040 values, serial MT19937) runs on the host while the GPU encodes; the copy goes from a
        pinned buffer on a side stream.  Same values, same position in the CPU generator's stream (nothing else draws from it
        in between).  -> (device tensor, event the consuming stream must wait for)."""
        samp_batch = B if constant_in_time else B * T
        size = (samp_batch, num_points, self.cnf_args.input_dim)
        # pinned buffer + copy stream per (device, size), at most two entries (evicting the oldest releases its pinned memory); kept in a
        # module-level table, not on the module: streams / events must not end up in copy.deepcopy(model) or torch.save(model)
        key = (id(self), str(device), size)
        ent = _EARLY_DRAW.get(key)
        if ent is None:
            mine = [k for k in _EARLY_DRAW if k[0] == id(self)]
            for k in mine[:max(0, len(mine) - 1)]:
                old = _EARLY_DRAW.pop(k)
                if old["ev"] is not None:
                    old["ev"].synchronize()
            ent = _EARLY_DRAW[key] = {"buf": torch.empty(size, dtype=torch.float32, pin_memory=True), "ev": None, "stream": torch.cuda.Stream(device=device)}
            weakref.finalize(self, _drop_early_draw, id(self))
        if ent["ev"] is not None:
            ent["ev"].synchronize()          # the previous call's copy out of the pinned buffer (a whole step ago)
        torch.randn(*size, out=ent["buf"])   # == torch.randn(*size): same generator, same stream position
        with torch.cuda.stream(ent["stream"]):
            yd = ent["buf"].to(device, non_blocking=True)
            # log N(y; 0, I) of the draw (caspr.py:258) right behind the copy, under the encoder (decode uses it unless it alters y)
            lp = standard_normal_logprob(yd).view(samp_batch, num_points, -1).sum(2)
            ent["ev"] = torch.cuda.Event()
            ent["ev"].record(ent["stream"])
        return yd, ent["ev"], lp

    def _base_samples(self, B, T, num_points, constant_in_time, truncate_std, sample_contours, y, like, early=None):
        """The base-distribution draw of decode (caspr.py:228-256) -> (B*T, num_points, 3) on `like`'s device."""
        samp_batch = B if constant_in_time else B * T
        input_dim = self.cnf_args.input_dim
        samp_size = (samp_batch, num_points, input_dim)
        if y is not None:
            y = y.to(like).reshape(B * T, num_points, input_dim)
            constant_in_time = False
        elif sample_contours is not None:
            radii = sample_contours
            contours = []
            nsamp_pts = 0
            for radius in radii:
                last = radius == radii[-1]
                cnt = (num_points - nsamp_pts) if last else (num_points // len(radii))
                pts = sphere_surface_points(samp_batch * cnt, radius=radius).reshape((samp_batch, cnt, 3))
                contours.append(pts)
                nsamp_pts += num_points // len(radii)
            y = torch.from_numpy(np.concatenate(contours, axis=1)).to(like).view(samp_size)
        elif early is not None:
            y, ev = early[0], early[1]
            torch.cuda.current_stream().wait_event(ev)
            y.record_stream(torch.cuda.current_stream())
            if truncate_std is not None:
                truncated_normal(y, mean=0, std=1, trunc_std=truncate_std)
        else:
            y = sample_gaussian(samp_size, truncate_std, device=like.device)
        if constant_in_time:
            y = y.view((B, 1, num_points, input_dim)).expand((B, T, num_points, input_dim)).reshape((B * T, num_points, input_dim))
        return y.contiguous()

    def decode(self, z, num_points=1024, constant_in_time=False, truncate_std=None, sample_contours=None, y=None, _early=None):
        """caspr.py:204-267.  `y` (B,T,num_points,3) optionally supplies the base samples."""
        B, T, H = z.size()
        input_dim = self.cnf_args.input_dim
        given = y is not None
        y = self._base_samples(B, T, num_points, constant_in_time, truncate_std, sample_contours, y, z, early=_early)
        if _early is not None and not given and sample_contours is None and truncate_std is None and not constant_in_time:
            logp_y = _early[2]                       # computed behind the draw's copy (same values: same op on the same tensor)
            logp_y.record_stream(torch.cuda.current_stream())
        else:
            logp_y = standard_normal_logprob(y).view(B * T, num_points, -1).sum(2)
        z = z.reshape((B * T, H))
        guard = self._guard_cnf_begin(y, z) if (self.check_tol is not None and y.is_cuda and not torch.is_grad_enabled()) else None
        x = self.point_cnf(y, z, reverse=True)
        if guard is not None:
            self._guard_cnf_end(guard, x)
        return y.view((B, T, num_points, input_dim)), logp_y.view((B, T, num_points)), x.view((B, T, num_points, input_dim))

    def reconstruct(self, x, num_points=1024, constant_in_time=False, timestamps=None, max_timestamp=5.0,
                    truncate_std=None, sample_contours=None, y=None, check_tol=_UNSET):
        """caspr.py:269-308 -> (y, logp_y, x, tnocs_pred).  `y` (B,T,num_points,3) optionally supplies the base samples.
        check_tol: the accuracy guard's tolerance for THIS call (default: the model's `check_tol` attribute; None = off)."""
        if check_tol is not _UNSET:
            prev, self.check_tol = self.check_tol, check_tol
            try:
                return self.reconstruct(x, num_points, constant_in_time, timestamps, max_timestamp, truncate_std, sample_contours, y)
            finally:
                self.check_tol = prev
        with torch.no_grad():
            B, T, N, _ = x.size()
            if timestamps is None:
                all_times = x[:, :, 0, 3] / max_timestamp
            else:
                all_times = timestamps.view((1, -1)).repeat((B, 1)).to(x)
            # the T-NOCS regression of the encoder's last layer runs on a side stream underneath the latent solve (32 workgroups,
            # latency-bound); joined before the flow starts
            defer = x.is_cuda and not self._differentiable(x)
            # what the latent solve needs from the time stamps alone is queued before the encoder (a dozen tiny kernels that
            # otherwise sit between the encoder's last layer and the solve, behind the T-NOCS layer's workgroups)
            plan, early_lat, st = None, None, None
            if defer:
                # (the time-stamp bookkeeping -- sort, inverse permutation, evaluation count: a dozen tiny kernels -- on the stream the
                # solve will run on, not in front of the encoder)
                st = _EARLY_STREAM.get(str(x.device))
                if st is None:
                    st = _EARLY_STREAM[str(x.device)] = torch.cuda.Stream(device=x.device)
                st.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(st):
                    plan = self.latent_ode.plan_times(all_times)
                    plan["event"] = torch.cuda.Event()
                    plan["event"].record()
                all_times.record_stream(st)
            # worth it while the reserved units cost the layer less than the solve takes: one team (<= 16 sequences = 32 units for the
            # two thirds of a layer that lasts ~5 ms per 327,680 rows) against ~2.3 ms of solve; larger batches / longer layers keep the
            # serial order (cfg-5: 4 teams would halve the chip under a 75 ms layer).  Either order gives the same bits.
            if defer and EARLY_LATENT and self.latent_ode.input_size <= 64 and B <= 16 and B * T * N <= 800000:
                early_lat = _EarlyLatent(self.latent_ode, plan, st)
            z0, tnocs_pred = self.encoder(x, defer_tnocs=True, early=early_lat) if defer else self.encode(x)
            # the encoder is queued: draw the base samples on the host now (as the reference does inside decode), under it
            early = self._draw_early(B, T, num_points, constant_in_time, x.device) if (defer and y is None and sample_contours is None) else None
            with ops.timed("latent"):
                z = self.aggregate_and_solve_latent(z0, all_times, plan, early_lat)
            self._early_latent_used = bool(early_lat is not None and early_lat.event is not None)     # for tests / tools
            if defer and not JOIN_TNOCS_LATE:
                self.encoder.join()
            if defer and JOIN_TNOCS_LATE:
                ops.BEFORE_CNF_LAUNCH = self.encoder.launch_tnocs        # queued between the flow's hyper conv and the flow (tpointnet2.py)
            try:
                with ops.timed("decode"):
                    y, logp_y, x = self.decode(z, num_points, constant_in_time, truncate_std, sample_contours, y=y, _early=early)
            finally:
                ops.BEFORE_CNF_LAUNCH = None
                if defer and JOIN_TNOCS_LATE:
                    # the T-NOCS regression (a 0.5 ms HBM-bound conv on the side stream) is joined BEHIND the flow's launch, not in front of
                    # it: nothing of the flow reads it, and its workgroups drain while the flow's first ones start.  In the `finally`: a
                    # CasprAccuracyError of an EARLIER call surfacing inside decode() must not leave the side stream unjoined
                    self.encoder.join()
            return y, logp_y, x, tnocs_pred

    def calibrate_rk4_steps(self, x, tol=1e-6, candidates=(1, 2, 4, 8, 16, 32, 64, 128), num_points=512, timestamps=None, max_timestamp=5.0,
                            latent_tol=None, latent_candidates=(1, 2, 4, 8, 16, 32), refine=True, rtol=0.0):
        """Pick the CNF's fixed RK4 step count the way an adaptive solver picks its step: by an error estimate on the
        actual weights and input.  Decodes every sequence of `x` with S and 2S steps (same base samples) for each
        candidate S in ascending order and keeps the first S whose error estimate (16/15) max|x_S - x_2S| (Richardson: the
        step-doubling difference is 15/16 of the S-step error of a 4th-order method) is <= tol; with `refine` (default) the counts BETWEEN the last failing candidate and
        that one are then tried where the S^-4 law predicts they pass (_first_passing: prediction verified by an actual S-vs-2S
        solve), so the result need not be a power of two.  Sets `rk4_steps` on every CNF block; returns (S, {S: difference}) with the
        differences of the counts that were tried.  `rtol` > 0 relaxes the bound to tol + rtol max|x| (torchdiffeq's atol + rtol |x|,
        the form the run-time guard uses with atol = rtol = check_tol); the default is the absolute bound.  With `latent_tol` the latent ODE's steps per interval are chosen FIRST, the
        same way (max|z_L - z_2L| <= latent_tol over the requested stamps), and installed on `latent_ode.rk4_steps`; the return
        value then is (S, diffs, L, latent_diffs).
        The reference's dopri5 runs at atol = rtol = 1e-5 (flow.py:96-99; 1e-3 for the latent ODE, latent_ode_model.py:38,83);
        the defaults of this build (8 steps, 2 per interval) are kept unless this is called."""
        from .cnf import CNF
        blocks = [l for l in self.point_cnf.chain if isinstance(l, CNF)]
        guard, self.check_tol = self.check_tol, None          # the candidates below are MEANT to be under-resolved: no guard while choosing
        try:
            return self._calibrate(blocks, x, tol, candidates, num_points, timestamps, max_timestamp, latent_tol, latent_candidates, refine, rtol)
        finally:
            self.check_tol = guard

    def _calibrate(self, blocks, x, tol, candidates, num_points, timestamps, max_timestamp, latent_tol, latent_candidates, refine=True, rtol=0.0):
        with torch.no_grad():
            xs = x                   # every sequence of x (round 4 looked at the first only: the worst one decides)
            z0, _ = self.encode(xs)
            times = xs[:, :, 0, 3] / max_timestamp if timestamps is None else timestamps.view(1, -1).repeat(xs.shape[0], 1).to(xs)
            lat = None
            if latent_tol is not None:
                zs = {}

                def zsol(L):
                    if L not in zs:
                        self.latent_ode.rk4_steps = L
                        zs[L] = self.aggregate_and_solve_latent(z0, times)
                    return zs[L]
                lchosen, ldiffs = _first_passing(lambda L: float((zsol(L) - zsol(2 * L)).abs().max()), latent_candidates, latent_tol * 15.0 / 16.0, refine)
                self.latent_ode.rk4_steps = lchosen
                lat = (lchosen, ldiffs)
            z = self.aggregate_and_solve_latent(z0, times)
            y = torch.randn(z.shape[0], z.shape[1], num_points, self.cnf_args.input_dim, device=x.device)
            sols = {}

            def sol(S):
                if S not in sols:
                    for b in blocks:
                        b.rk4_steps = S
                    sols[S] = self.decode(z, num_points, y=y)[2]
                return sols[S]
            bound = tol + rtol * float(sol(max(candidates)).abs().max()) if rtol > 0 else tol
            bound *= 15.0 / 16.0       # the criterion is on the ERROR of the S-step solution, (16 / 15) max |x_S - x_2S|, not on the difference
            chosen, diffs = _first_passing(lambda S: float((sol(S) - sol(2 * S)).abs().max()), candidates, bound, refine)
        for b in blocks:
            b.rk4_steps = chosen
        self.cnf_args.rk4_steps = chosen
        return (chosen, diffs) if lat is None else (chosen, diffs, lat[0], lat[1])
