"""cfg-3 (BASELINE.json configs[2]: cars.cfg training, T=10 N=1024, batch 64 = 8 sequences per GPU) AT ITS OWN SIZE:
one rank's shard (8, 10, 1024) through `model.train(); losses = model(x, sp); loss.backward()` -- the body of
run_one_epoch (train_utils.py:120-176).  The small-shape tests of tests/test_hip_train.py pin every gradient kernel
against f64 autograd; what only this size reaches is the 60 GB activation tape, the 4096-row weight-gradient slabs and
the segment CSR of the scatter-adds at 81,920 rows.  Checked here:
  * the loss of one sequence against the CPU oracle's differentiable mode (itself pinned to the real reference's
    captured training step, tests/test_oracle_golden.py) and the gradient of the last encoder layer against it;
  * shard identity: the gradient over the 8-sequence batch == the mean of the 8 single-sequence gradients
    (what 8 ranks with one sequence each + the gradient all-reduce would produce);
  * bit-reproducibility of the whole step; a finite gradient for every trainable tensor; peak memory, recorded.
"""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
B, T, N = 8, 10, 1024
NON_PARAM = ("running_mean", "running_var", "step", "_num_evals")


def _report(d):
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "cfg3_train_report.json"), "w") as f:
        json.dump(d, f, indent=1, sort_keys=True)


def test_cfg3_training_step_at_its_own_size(seeded_sd):
    from caspr_amd.models import CaSPR
    from caspr_amd.train.loop import training_loss
    from caspr_amd.utils.synthetic import car_sequences
    from oracle import model as O
    dev = torch.device("cuda:0")
    x, sp = car_sequences(B, T, N, seed=303)
    e = torch.from_numpy(np.random.default_rng(304).normal(0, 1, (B * T, N, 3)).astype(np.float32))
    xd, spd, ed = x.to(dev), sp.to(dev), e.to(dev)
    m = CaSPR()                                    # cars.cfg: every model option at its default, RK4 8 / 2 steps
    m.load_state_dict(seeded_sd)
    m = m.to(dev).train()
    stats = {k: v.clone() for k, v in m.state_dict().items() if "running_" in k or k.endswith(".step")}
    names = [n for n, _ in m.named_parameters()]

    def step(xs, sps, es):
        m.load_state_dict(stats, strict=False)     # MovingBatchNorm statistics move at every training-mode call: rewind them
        m.zero_grad(set_to_none=True)
        loss, _, _ = training_loss(m(xs, sps, e=es), 0.01, 100.0)
        loss.backward()
        return float(loss.detach()), [p.grad.detach().clone() for p in m.parameters()]

    torch.cuda.reset_peak_memory_stats()
    l_all, g_all = step(xd, spd, ed)
    peak_gb = torch.cuda.max_memory_allocated() / 2 ** 30
    rep = {"shape": [B, T, N], "loss": l_all, "peak_mem_GB": round(peak_gb, 1), "trainable_tensors": len(names)}
    _report(rep)
    # every trainable tensor has a finite gradient (the duplicated latent_ode.solver.ode_func.* parameters alias latent_ode.ode_func.*)
    assert len(g_all) == len(names) and len(names) >= 221, len(names)
    bad = [n for n, g in zip(names, g_all) if g is None or not bool(torch.isfinite(g).all())]
    assert not bad, "non-finite / missing gradients: %s" % bad[:5]
    assert np.isfinite(l_all)

    # ---- bit-reproducibility at this size (fixed-order slab combines, segment gathers)
    l_rep, g_rep = step(xd, spd, ed)
    diff = [n for n, u, v in zip(names, g_all, g_rep) if not torch.equal(u, v)]
    assert l_rep == l_all and not diff, "step not bit-reproducible: %s" % diff[:5]

    # ---- shard identity: batch of 8 == mean of 8 singles
    acc = [torch.zeros_like(g, dtype=torch.float64) for g in g_all]
    l_sum, l0, g0 = 0.0, None, None
    for b in range(B):
        l_b, g_b = step(xd[b:b + 1], spd[b:b + 1], ed[b * T:(b + 1) * T])
        l_sum += l_b
        for a_, g in zip(acc, g_b):
            a_ += g.double()
        if b == 0:
            l0, g0 = l_b, g_b
    # A conv bias in front of a GroupNorm with ONE channel per group (the two 16-wide layers of the first set-abstraction scale:
    # GroupNorm(16, 16), pointnet2.py:649-703) cannot change the output: its gradient is mathematically zero, and what any f32
    # implementation returns for it -- the reference's autograd included -- is the rounding noise of a sum of 1.3 M terms of
    # size 1/sigma.  Those two tensors are left out of the identity (their magnitude is recorded).
    zero_grad = {"encoder.local_extract.set_abstractions.0.pointnet_modules.0.conv_layers.%d.bias" % i for i in (0, 1)}
    num = den = 0.0
    worst = ("", 0.0)
    for n, ga, a_ in zip(names, g_all, acc):
        avg = a_ / B
        if n in zero_grad:
            rep["zero_gradient_noise_l2:" + n] = [float(ga.double().norm()), float(avg.norm())]
            continue
        d, r = float((ga.double() - avg).norm()), float(avg.norm())
        num, den = num + d * d, den + r * r
        if d > worst[1]:
            worst = (n, d)
    rep.update({"shard_identity_rel_l2": (num / den) ** 0.5, "shard_identity_worst": worst[0], "shard_identity_worst_abs_l2": worst[1],
                "grad_l2": den ** 0.5, "loss_mean_of_singles": l_sum / B})
    _report(rep)
    assert abs(l_all - l_sum / B) <= 1e-5 * abs(l_all), (l_all, l_sum / B)
    # same kernels on the same per-sequence data: only the f32 reduction order across the batch differs (weight-gradient slabs,
    # GroupNorm beta / gamma sums over 1.3 M gathered rows with heavy cancellation): measured 4e-5 of the gradient's norm, the
    # worst tensor (the first GroupNorm bias of the 16-wide scale) 0.04 of 1006
    assert (num / den) ** 0.5 <= 2e-4 and worst[1] <= 2e-4 * den ** 0.5, rep

    # ---- sequence 0 against the CPU oracle's differentiable mode (f32 torch-CPU autograd + the C point ops)
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    s_ = {k: (v.detach().clone().requires_grad_(True) if v.is_floating_point() and not k.endswith(NON_PARAM) else v) for k, v in seeded_sd.items()}
    want, _, _ = O.training_loss(s_, x[:1], sp[:1], e[:T], cnf_steps=m.cnf_args.rk4_steps, latent_steps=m.latent_ode.rk4_steps)
    want.backward()
    w = float(want.detach())
    gi = names.index("encoder.conv3.weight")
    gw = s_["encoder.conv3.weight"].grad
    e_c3 = float((g0[gi].cpu() - gw).norm() / gw.norm())
    gi2 = names.index("point_cnf.chain.1.odefunc.diffeq.layers.3._layer.weight")
    gw2 = s_["point_cnf.chain.1.odefunc.diffeq.layers.3._layer.weight"].grad
    e_l3 = float((g0[gi2].cpu() - gw2).norm() / gw2.norm())
    rep.update({"seq0_loss_hip": l0, "seq0_loss_oracle": w, "seq0_conv3_grad_rel_l2": e_c3, "seq0_cnf_out_layer_grad_rel_l2": e_l3})
    _report(rep)
    assert abs(l0 - w) <= 2e-5 * abs(w), (l0, w)
    assert e_c3 <= 1e-4 and e_l3 <= 1e-3, (e_c3, e_l3)
