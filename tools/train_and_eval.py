#!/usr/bin/env python
"""End to end on the GPU box: TRAIN the model with the HIP training tier on the synthetic rigid-car distribution (fresh sequences every
step, the reference's loss and optimiser: train_utils.py:151-176, train.py:135), write the checkpoint in the reference's format
(train.py:176-188: `torch.save(model.state_dict(), ...)`), load it back the way test.py:104-107 / `bench.py --weights` do, and look at the
TRAINED weights with everything the build has for the seeded ones:

  * held-out reconstruction quality before / after training (Chamfer-L2 x1000 against the ground-truth NOCS points, T-NOCS regression
    error: utils/evaluations.py, the reference's protocol at 10 x 2048);
  * how hard the trained flow is to integrate: the step-doubling calibration (`calibrate_rk4_steps`, tol 1e-5), the run-time guard's
    estimate at the default 8 / 2 steps, and what the reference's own integrator does on these weights -- the oracle's dopri5
    (rtol = atol = 1e-5, flow.py:96-99) in f64: its evaluation count and its distance from the HIP result;
  * parity on the trained weights: HIP reconstruct() against the f64 oracle (same discrete RK4 map) on a held-out sequence, flat 1e-5;
  * throughput of the headline call on the trained weights (same kernels, for the record).

There is no network: the pretrained `caspr_weights_cars.pth` of BASELINE.json's configs[1] cannot be fetched; this is the closest thing
the box can produce -- a checkpoint that went through the whole surface (train -> save -> load -> reconstruct -> evaluate).

    python tools/train_and_eval.py --steps 300 --out gpurun_out/trained_report.json         (GPU; ~2 min at 300 steps)
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch


def main(argv=None, timed_steps=None):
    """-> the report (also written to --out).  timed_steps(k, step) -> (seconds, outputs): bench.py's own bracketed clock for the headline
    legs (default: perf_counter around synchronize, as below)."""
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--seq-len", type=int, default=10)
    ap.add_argument("--num-pts", type=int, default=1024)
    ap.add_argument("--lr", type=float, default=1e-4, help="train.py's default (cars.cfg)")
    ap.add_argument("--eval-seqs", type=int, default=4)
    ap.add_argument("--ckpt", default="/tmp/caspr_amd_trained/time_model_0.pth")
    ap.add_argument("--out", default=None)
    ap.add_argument("--no-oracle", action="store_true", help="skip the CPU f64 oracle legs (parity, dopri5)")
    ap.add_argument("--no-dopri5", action="store_true", help="skip the oracle's dopri5 leg only")
    ap.add_argument("--no-headline", action="store_true", help="skip timing the headline call on the trained weights")
    ap.add_argument("--probe-at", default="", metavar="N1,N2", help="training steps at which the model is calibrated (tol 1e-5) and the headline call timed at the "
                    "calibrated counts, WITHOUT disturbing the run (step counts and both generators' states restored): how the flow's stiffness grows with training")
    args = ap.parse_args(argv)

    from caspr_amd import ops
    from caspr_amd.models import CaSPR
    from caspr_amd.train.loop import train_step
    from caspr_amd.utils.synthetic import car_sequences, seeded_state_dict
    from caspr_amd.utils.torch_utils import load_weights
    from caspr_amd.utils import evaluations as E

    dev = torch.device("cuda:0")
    rep = {"what": "train (HIP tier) -> checkpoint (reference format) -> load -> reconstruct / evaluate, synthetic rigid cars",
           "train": {"steps": args.steps, "batch": args.batch, "seq_len": args.seq_len, "num_pts": args.num_pts, "lr": args.lr,
                     "cnf_loss_weight": 0.01, "tnocs_loss_weight": 100.0, "data": "caspr_amd.utils.synthetic.car_sequences, a fresh seed per step"}}
    model = CaSPR()
    model.load_state_dict(seeded_state_dict(model.state_dict(), 0))
    model = model.to(dev)

    T, N_eval = 10, 2048
    xe, spe = car_sequences(args.eval_seqs, T, N_eval, seed=777000)          # held out: seeds the training never draws
    torch.manual_seed(99)
    y_eval = torch.randn(args.eval_seqs, T, N_eval, 3)

    def evaluate(m):
        m.eval()
        with torch.no_grad():
            r = E.test_shape_recon(m, [(xe, spe)], dev, base_samples=[y_eval])
            t = E.test_tnocs_regression(m, [(xe, spe)], dev)
        ops.check_deferred_errors()
        return {"chamfer_x1000": r["observed_chamfer_x1000"], "emd_x1000": r["observed_emd_x1000"], "tnocs_space_l2": t["space"], "tnocs_time_l1": t["time"]}

    rep["held_out_before"] = evaluate(model)
    print("before training:", json.dumps(rep["held_out_before"]), flush=True)

    # ---- train
    opt = torch.optim.Adam(model.parameters(), lr=args.lr, betas=(0.9, 0.999), eps=1e-8)
    curve = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    probes = sorted(int(v) for v in args.probe_at.split(",") if v.strip())
    along = []
    xb_p = None

    def probe(at):
        """Calibrate the CURRENT weights and time the headline call at the calibrated counts; the training run continues as if this had not happened."""
        nonlocal xb_p
        from caspr_amd.models.cnf import CNF as _CNF
        cpu_rng, gpu_rng = torch.get_rng_state(), torch.cuda.get_rng_state(dev)
        blocks = [b_ for b_ in model.point_cnf.chain if isinstance(b_, _CNF)]
        keep = ([b_.rk4_steps for b_ in blocks], model.cnf_args.rk4_steps, model.latent_ode.rk4_steps, model.training)
        model.eval()
        try:
            with torch.no_grad():
                S_, d_, L_, ld_ = model.calibrate_rk4_steps(xe.to(dev), tol=1e-5, num_points=512, timestamps=spe[0, :, 0, 3].to(dev), latent_tol=1e-5)
                if xb_p is None:
                    xb_, spb_ = car_sequences(16, 10, 2048, seed=1234)
                    xb_p = (xb_.to(dev), spb_[0, :, 0, 3].to(dev))
                import warnings
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    model.reconstruct(xb_p[0], num_points=2048, timestamps=xb_p[1])
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    for _ in range(4):
                        model.reconstruct(xb_p[0], num_points=2048, timestamps=xb_p[1])
                    torch.cuda.synchronize()
                    ms_ = (time.perf_counter() - t1) / 4 * 1e3
                    ops.check_deferred_errors()
            along.append({"train_steps": at, "cnf_rk4_steps": S_, "latent_rk4_steps": L_, "cnf_step_doubling_diffs": {str(k): v for k, v in d_.items()},
                          "ms_per_step": round(ms_, 3), "sequences_per_sec": round(16e3 / ms_, 2)})
        finally:
            for b_, st_ in zip(blocks, keep[0]):
                b_.rk4_steps = st_
            model.cnf_args.rk4_steps, model.latent_ode.rk4_steps = keep[1], keep[2]
            model.train(keep[3])
            torch.set_rng_state(cpu_rng)
            torch.cuda.set_rng_state(gpu_rng, dev)
            ops.reset_guard()

    for step in range(args.steps):
        if step in probes:
            probe(step)
        x, sp = car_sequences(args.batch, args.seq_len, args.num_pts, seed=100000 + step * args.batch)
        loss, cnf_l, tnocs_l = train_step(model, opt, x.to(dev), sp.to(dev))
        if step % 10 == 0 or step == args.steps - 1:
            curve.append({"step": step, "loss": loss, "nll_per_point_sum": cnf_l, "tnocs_l1": tnocs_l})
            print("step %4d  loss %.5f  (cnf %.5f, tnocs %.6f)" % (step, loss, cnf_l, tnocs_l), flush=True)
    torch.cuda.synchronize()
    rep["train"]["wall_s"] = round(time.perf_counter() - t0, 2)
    rep["train"]["curve"] = curve
    rep["train"]["finite"] = all(c["loss"] == c["loss"] and abs(c["loss"]) < 1e30 for c in curve)
    rep["along_training"] = along
    ops.check_deferred_errors()

    # ---- checkpoint in the reference's format, loaded back as test.py / bench.py --weights load one
    os.makedirs(os.path.dirname(args.ckpt), exist_ok=True)
    torch.save(model.state_dict(), args.ckpt)
    trained = CaSPR()
    ck = torch.load(args.ckpt, map_location="cpu")
    load_weights(trained, ck)
    trained = trained.to(dev).eval()
    a, b = model.state_dict(), trained.state_dict()
    rep["checkpoint"] = {"file": os.path.basename(args.ckpt), "bytes": os.path.getsize(args.ckpt), "keys": len(ck),
                         "round_trip_bitwise": bool(set(a) == set(b) and all(torch.equal(a[k].cpu(), b[k].cpu()) for k in a))}
    rep["held_out_after"] = evaluate(trained)
    print("after training: ", json.dumps(rep["held_out_after"]), flush=True)

    # ---- how hard is the trained flow to integrate?
    xg = xe.to(dev)
    ts = spe[0, :, 0, 3].to(dev)
    with torch.no_grad():
        S, diffs, L, ldiffs = trained.calibrate_rk4_steps(xg, tol=1e-5, num_points=512, timestamps=ts, latent_tol=1e-5)
    rep["calibration_tol_1e-5"] = {"cnf_rk4_steps": S, "cnf_step_doubling_diffs": {str(k): v for k, v in diffs.items()},
                                   "latent_rk4_steps": L, "latent_step_doubling_diffs": {str(k): v for k, v in ldiffs.items()}}
    # back to the defaults the bench runs (8 CNF steps, 2 per latent interval), with the guard on
    from caspr_amd.models.cnf import CNF
    for blk in trained.point_cnf.chain:
        if isinstance(blk, CNF):
            blk.rk4_steps = 8
    trained.cnf_args.rk4_steps = 8
    trained.latent_ode.rk4_steps = 2
    guard = {"check_tol": 1e-5}
    ops.reset_guard()
    try:
        import warnings
        with warnings.catch_warnings(record=True) as wrec:
            warnings.simplefilter("always")
            with torch.no_grad():
                trained.reconstruct(xg, num_points=N_eval, timestamps=ts, y=y_eval.to(dev), check_tol=1e-5)
            torch.cuda.synchronize()
            ops.check_deferred_errors()
        guard["verdict"] = "quiet" if not wrec else "warned: " + "; ".join(str(w.message)[:200] for w in wrec)
    except Exception as ex:          # CasprAccuracyError when the guard's action is "raise"
        guard["verdict"] = "raised: " + str(ex)[:300]
    guard["estimates"] = {k: {"estimate": v.get("estimate"), "bound": v.get("bound"), "steps": v.get("steps"), "other_steps": v.get("other_steps")}
                          for k, v in ops.GUARD_LAST.items()}
    rep["guard_at_8_and_2_steps"] = guard

    def set_steps(S_, L_):
        for blk in trained.point_cnf.chain:
            if isinstance(blk, CNF):
                blk.rk4_steps = S_
        trained.cnf_args.rk4_steps = S_
        trained.latent_ode.rk4_steps = L_

    # ---- parity and the reference's integrator on the trained weights (CPU, f64): at the default 8 / 2 steps and at the CALIBRATED counts
    if not args.no_oracle:
        from oracle import model as O
        sd64 = {k: v.detach().cpu().double() for k, v in trained.state_dict().items()}
        x1, y1, ts1 = xe[:1], y_eval[:1], spe[0, :, 0, 3]
        torch.set_num_threads(min(32, os.cpu_count() or 1))
        rep["parity_trained_weights"] = {"sequence": "held-out 0, (1, 10, 2048), given base samples", "oracle_seconds": {}}
        x64_by = {}
        for name, (S_, L_) in (("hip_vs_f64_oracle_same_rk4_map", (8, 2)), ("hip_vs_f64_oracle_at_calibrated_steps", (S, L))):
            if (S_, L_) in x64_by:
                rep["parity_trained_weights"][name] = dict(rep["parity_trained_weights"]["hip_vs_f64_oracle_same_rk4_map"])
                continue
            set_steps(S_, L_)
            with torch.no_grad():
                got = trained.reconstruct(x1.to(dev), num_points=N_eval, timestamps=ts1.to(dev), y=y1.to(dev), check_tol=None)
            gx, gt = got[2].cpu().double(), got[3].cpu().double()
            t1 = time.perf_counter()
            _, _, x64, t64 = O.reconstruct(sd64, x1.double(), y1.double(), timestamps=ts1.double(), cnf_steps=S_, latent_steps=L_)
            rep["parity_trained_weights"]["oracle_seconds"]["rk4_%d_%d" % (S_, L_)] = round(time.perf_counter() - t1, 1)
            x64_by[(S_, L_)] = (gx, x64)
            ex_, et_ = float((gx - x64).abs().max()), float((gt - t64).abs().max())
            rep["parity_trained_weights"][name] = {"cnf_rk4_steps": S_, "latent_rk4_steps": L_, "x": ex_, "tnocs": et_, "bound": 1e-5,
                                                   "ok": bool(ex_ <= 1e-5 and et_ <= 1e-5)}
        set_steps(8, 2)
        gx, x64 = x64_by[(8, 2)]
        rep["parity_trained_weights"]["hip_nfe"] = [int(v) for v in trained.get_nfe()]
        if not args.no_dopri5:
            nfe = [0, 0]
            t1 = time.perf_counter()
            _, _, xd5, _ = O.reconstruct(sd64, x1.double(), y1.double(), timestamps=ts1.double(), method="dopri5", nfe=nfe)
            rep["parity_trained_weights"]["oracle_seconds"]["dopri5"] = round(time.perf_counter() - t1, 1)
            rep["parity_trained_weights"]["reference_integrator_dopri5_f64"] = {
                "rtol_atol": 1e-5, "nfe_latent": int(nfe[0]), "nfe_cnf": int(nfe[1]), "hip_rk4_vs_dopri5_x": float((gx - xd5).abs().max()),
                "f64_rk4_vs_dopri5_x": float((x64 - xd5).abs().max())}
        print("parity:", json.dumps(rep["parity_trained_weights"]), flush=True)

    if args.no_headline:
        if args.out:
            with open(args.out, "w") as f:
                json.dump(rep, f, indent=1)
        return rep
    # ---- the headline call on the trained weights (the model's DEFAULTS: run-time guard on, reporting as a warning), at the default
    # 8 / 2 steps and at the counts the calibration chose for THESE weights (what "within 1e-5 of the converged solution" costs here)
    xb, spb = car_sequences(16, 10, 2048, seed=1234)
    xb, tsb = xb.to(dev), spb[0, :, 0, 3].to(dev)

    def own_clock(k, step):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        o = None
        for _ in range(k):
            o = step()
        torch.cuda.synchronize()
        return time.perf_counter() - t1, o
    clock = timed_steps or own_clock

    def headline():
        with torch.no_grad():
            return trained.reconstruct(xb, num_points=2048, timestamps=tsb)

    def timed(S_, L_, k):
        import warnings
        set_steps(S_, L_)
        ops.check_deferred_errors()
        ops.reset_guard()
        with warnings.catch_warnings(record=True) as wrec:
            warnings.simplefilter("always")
            headline()
            el, _ = clock(k, headline)
            torch.cuda.synchronize()
            ops.check_deferred_errors()
        spoke = [str(w.message)[:160] for w in wrec if "not converged" in str(w.message)]
        return {"cnf_rk4_steps": S_, "latent_rk4_steps": L_, "steps": k, "ms_per_step": round(1e3 * el / k, 3), "sequences_per_sec": round(16 * k / el, 2),
                "guard": {"check_tol": trained.check_tol, "verdict": "quiet" if not spoke else "warned (%d): %s" % (len(spoke), spoke[0]),
                          "worst_estimate_over_bound": round(ops.GUARD_HISTORY_MAX, 4),
                          "estimates": {k_: v.get("estimate") for k_, v in ops.GUARD_LAST.items()}}}
    rep["headline_on_trained_weights"] = dict(timed(8, 2, 5), workload="reconstruct(), B=16, T=10, N=2048, guard on (defaults)")
    rep["headline_on_trained_weights"]["at_calibrated_steps"] = timed(S, L, 5) if (S, L) != (8, 2) else dict(rep["headline_on_trained_weights"])
    set_steps(S, L)
    print(json.dumps(rep["headline_on_trained_weights"]), flush=True)
    if args.out:
        with open(args.out, "w") as f:
            json.dump(rep, f, indent=1)
    return rep


if __name__ == "__main__":
    main()
