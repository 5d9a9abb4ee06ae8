#!/usr/bin/env python
"""bench.py -- sequences/sec of CaSPR.reconstruct (encode -> latent advect -> CNF sample) on MI355X.

    python bench.py --gpus N --steps K --warmup W            (starts its own N ranks when N > 1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one reconstruct() pass over one batch of synthetic sequences already resident in HBM.
Workload (BASELINE.json configs[1]): cars.cfg rigid reconstruction, B=16 sequences per GPU, T=10, N=2048,
num_points=2048, all steps observed (evaluations.py:111-114), fixed-step RK4 (8 CNF steps = 32 function evaluations,
2 latent RK4 steps per interval).  `--clouds random --batch 64 --seq-len 20 --num-pts 4096` is configs[4].
Weak scaling: every rank owns its own B sequences (sequences are independent) -- no data-path
collective; value = all ranks' sequences / max-over-ranks time.

Arithmetic: f32 throughout.  The matrix products run on the bf16x6 kernels by default (each f32 operand split exactly
into three bf16 numbers, six bf16-MFMA partial products, f32 accumulation: f32-equivalent results, see caspr_amd/ops.py);
the same step on the pure f32-MFMA kernels is timed in the same process and reported as the `f32_mfma_path` sub-block.

Rank 0 prints ONE JSON line with the contract fields plus
  roofline      : the dominant kernel (the CNF solve) -- algorithmic FLOPs per launch / mean launch duration measured with
                  HIP events on the launch stream inside the timed region, vs the peak of the pipe it runs on;
  cpu_baseline  : the CPU oracle (a port: the reference's own CPU path cannot run) timed on this box's
                  host cores on 2 sequences of the same workload, and the ASSERTED parity of the HIP path against it
                  (a failed parity check still prints the line, with "parity_ok": false, and exits non-zero).
"""
import argparse
import json
import os
import platform
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_MFMA_F32_TFLOPS = 157.3            # /opt/skills/guides/MI355X_MICROARCH.md: dense f32-input MFMA peak
PEAK_MFMA_BF16_TFLOPS = 2500.0          # same guide: dense bf16 MFMA peak; a bf16x6 product costs six of them -> / 6
CNF_FLOP_PER_POINT_EVAL = 2 * (3 * 512 + 512 * 512 + 512 * 512 + 512 * 3)   # 1,054,720 (no divergence)


def cpu_description():
    model = platform.processor() or "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return model, os.cpu_count() or 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=16, help="sequences per GPU")
    ap.add_argument("--seq-len", type=int, default=10)
    ap.add_argument("--num-pts", type=int, default=2048)
    ap.add_argument("--cnf-steps", type=int, default=8)
    ap.add_argument("--latent-steps", type=int, default=2)
    ap.add_argument("--clouds", choices=["cars", "random"], default="cars",
                    help="cars: rotating box-surface clouds (configs[1]); random: i.i.d. U(0,1)^3 clouds (configs[4])")
    ap.add_argument("--matmul", choices=["bf16x6", "f32"], default=None, help="kernels of the matrix products (default: bf16x6)")
    ap.add_argument("--weights", default=None, metavar="PATH",
                    help="checkpoint with the reference's key surface (caspr_weights_cars.pth, test.py:104-107); default: seeded random init")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-f32-subblock", action="store_true", help="skip timing the same step on the f32-MFMA kernels")
    ap.add_argument("--no-guard-subblock", action="store_true", help="skip timing the same step with the run-time accuracy guard on")
    ap.add_argument("--no-sub-blocks", action="store_true",
                    help="skip the cfg5 / train_cfg3 / stress_dynamics sub-blocks (they run on one GPU only, after the timed region)")
    ap.add_argument("--no-train-subblock", action="store_true", help="N > 1: skip the sharded cfg-3 training step (the one with the gradient all-reduce)")
    ap.add_argument("--train-shape", default="8,10,1024", metavar="B,T,N", help="per-rank shape of that training step (tests shrink it)")
    ap.add_argument("--trained-steps", type=int, default=2000,
                    help="training steps of the trained_checkpoint sub-block (2000 = 6 minutes: where the default 8 CNF steps stop meeting 1e-5)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="TEST MODE (tests/test_multi_gpu.py on a 1-GPU box): the N ranks share the visible device(s) round-robin and talk over "
                         "gloo; exercises the rank logic of this file end to end, measures nothing -- the line carries value = null and says so")
    ap.add_argument("--calibrate-cnf-steps", type=float, default=None, metavar="TOL",
                    help="choose the CNF and latent step counts by step doubling at this tolerance (CaSPR.calibrate_rk4_steps) instead of "
                         "--cnf-steps / --latent-steps; default: 1e-5 with --weights (a trained flow), off (0) on the seeded weights, whose "
                         "headline number is quoted at the fixed, conservative 8 / 2 steps")
    args = ap.parse_args()

    # one process per GPU: a plain `python bench.py --gpus N` starts its own N ranks (torch.distributed.run on 127.0.0.1);
    # under an external launcher WORLD_SIZE must equal --gpus
    from caspr_amd.utils.launch import ensure_ranks
    rank, local_rank, world = ensure_ranks(args.gpus, __file__, sys.argv[1:], device_count=None if args.share_gpu else torch.cuda.device_count)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("gloo" if args.share_gpu else "nccl")   # "nccl" = RCCL on ROCm
    assert torch.cuda.is_available(), "bench.py needs a ROCm GPU (there is no CPU execution path)"
    if args.share_gpu:
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from caspr_amd import ops
    from caspr_amd.models import CaSPR
    from caspr_amd.utils.synthetic import seeded_state_dict, car_sequences, random_clouds, dense_sequences
    from caspr_amd.utils.sharding import max_over_ranks, shard_range, per_rank_values, collective_library
    from caspr_amd.utils.torch_utils import load_weights

    if args.matmul is not None:
        ops.set_matmul_mode(args.matmul)
    B, T, N = args.batch, args.seq_len, args.num_pts
    model = CaSPR(cnf_rk4_steps=args.cnf_steps, latent_rk4_steps=args.latent_steps)
    if args.weights:
        ck = torch.load(args.weights, map_location="cpu")
        load_weights(model, ck["model"] if isinstance(ck, dict) and "model" in ck and not torch.is_tensor(ck["model"]) else ck)
        weights_desc = "checkpoint %s" % os.path.basename(args.weights)
    else:
        model.load_state_dict(seeded_state_dict(model.state_dict(), 0))   # random-init weights of the architecture (no checkpoint offline)
        weights_desc = "seeded random-init weights"
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    model = model.to(dev).eval()

    # global batch = world * B sequences; this rank owns a contiguous block (weak scaling)
    lo, hi = shard_range(world * B, rank, world)
    if args.clouds == "cars":
        x_all, sp_all = car_sequences(hi - lo, T, N, seed=1234 + lo)
    else:
        x_all, sp_all = random_clouds(hi - lo, T, N, seed=1234 + lo), None
    # normalised time stamps = sample_points[..., 3] (caspr_dataset.py:200-204)
    times_cpu = sp_all[0, :, 0, 3].clone() if sp_all is not None else x_all[0, :, 0, 3] / 5.0
    x = x_all.to(dev)
    ts = times_cpu.to(dev)
    torch.manual_seed(rank)      # the base samples are drawn inside every reconstruct() call, on the CPU generator (models/utils.py:25)

    calibration = None
    if args.calibrate_cnf_steps is None:
        # a checkpoint that is not the seeded one: its flow is as hard to integrate as its training made it -- choose the step counts by
        # the error estimate (1e-5, the reference's tolerance), as an adaptive solver would; seeded weights: the fixed, conservative 8 / 2
        args.calibrate_cnf_steps = 1e-5 if args.weights else 0.0
    if args.calibrate_cnf_steps > 0:
        args.cnf_steps, diffs, args.latent_steps, ldiffs = model.calibrate_rk4_steps(x, tol=args.calibrate_cnf_steps, timestamps=ts,
                                                                                     latent_tol=args.calibrate_cnf_steps)
        calibration = {"tol": args.calibrate_cnf_steps, "chosen": args.cnf_steps, "step_doubling_diffs": {str(k): v for k, v in diffs.items()},
                       "latent_chosen": args.latent_steps, "latent_step_doubling_diffs": {str(k): v for k, v in ldiffs.items()}}

    def step():
        # the reference's own call (evaluations.py:108-114): the CPU draw of the base samples and their host-to-device copy are
        # INSIDE the step (reconstruct() overlaps them with the encoder, caspr_amd/models/caspr.py:_draw_early)
        with torch.no_grad():
            return model.reconstruct(x, num_points=N, timestamps=ts)

    rank_seconds = []

    def timed_steps(k, step=step):
        """k steps bracketed by barrier + synchronize on both sides; returns (seconds [max over ranks], last outputs)."""
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        ops.TIMERS.clear()
        ops.TIMING = True
        ops.TIMING_ONLY = {"cnf_rk4"}        # inside the timed region: the dominant kernel's event pair only (roofline.launch_ms);
                                             # the stage breakdown comes from the detail pass below (14 event records per step
                                             # cost 0.15 ms of the step)
        t0 = time.perf_counter()
        out_ = None
        for _ in range(k):
            out_ = step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        ops.TIMING = False
        rank_seconds[:] = per_rank_values(el, dev)
        return max_over_ranks(el, dev), out_

    import warnings as _warnings
    guard_warnings = []
    ops.reset_guard()
    with _warnings.catch_warnings(record=True) as _wrec:
        _warnings.simplefilter("always")
        for _ in range(args.warmup):
            step()
        elapsed, out = timed_steps(args.steps)
    for w_ in _wrec:
        if "not converged" in str(w_.message):
            guard_warnings.append(str(w_.message)[:200])
        else:
            _warnings.warn_explicit(w_.message, w_.category, w_.filename, w_.lineno)
    per_rank_ms = [round(1e3 * s_ / args.steps, 3) for s_ in rank_seconds]
    timers = {k: list(v) for k, v in ops.TIMERS.items()}
    mode = ops.matmul_mode()
    # ---- detail pass (NOT part of the headline timing): one HIP-event pair per launch of the other matrix kernels, for the
    # per-kernel roofline entries (every rank runs it so that the barriers of the sub-blocks below still pair up)
    torch.cuda.synchronize()
    ops.TIMERS.clear()
    ops.TIMING = 2
    # (per-kernel durations are measured WITHOUT the accuracy guard's check solve: queued behind a step's flow it runs under the next step's first
    # millisecond, beside the global PointNet's convs, whose event pairs would otherwise time the contention instead of the kernel -- the 128 -> 1024
    # layer read 2.2 ms instead of 1.0; the headline above and its own cnf_rk4 launch times are measured with the guard on)
    # ... and with the global PointNet on the caller's stream: on its own stream (the default schedule) its convs run beside the first
    # set-abstraction kernels since round 6 (the index chain in front of those got 0.5 ms shorter), and an event pair around a launch then times the
    # contention: 1.7 ms for the 128 -> 1024 layer against 1.0 ms beside the latency-bound index kernels alone.  The kernel table is about kernels;
    # the STAGE clocks below come from a second pass on the default schedule with the guard on, as the headline ran.
    import caspr_amd.models.tpointnet2 as _tp
    guard_tol, model.check_tol = model.check_tol, None
    gstream_on, _tp.GLOBAL_STREAM = _tp.GLOBAL_STREAM, False
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    model.check_tol, _tp.GLOBAL_STREAM = guard_tol, gstream_on
    ops.TIMING = False
    detail = {k: [a.elapsed_time(b_) for a, b_ in v] for k, v in ops.TIMERS.items() if k.startswith("k:")}
    ops.TIMERS.clear()
    ops.TIMING, ops.TIMING_ONLY = 1, None
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    ops.TIMING = False
    stage_timers = {k: list(v) for k, v in ops.TIMERS.items() if not k.startswith("k:")}      # the stage clocks of two steps on the headline's own schedule

    # ---- the same step on the pure f32-MFMA kernels (sub-block; every rank takes part so the barriers pair up)
    f32_block = None
    if not args.no_f32_subblock and (ops.CONV_BF16X6 or ops.CNF_BF16X6):
        prev = ops.set_matmul_mode("f32")
        step()
        k32 = max(1, min(args.steps, 3))
        el32, _ = timed_steps(k32)
        ev32 = ops.TIMERS.get("cnf_rk4", [])
        cnf32 = sum(a.elapsed_time(b) for a, b in ev32) / max(len(ev32), 1)
        flop32 = float((hi - lo) * T * N) * 4 * args.cnf_steps * CNF_FLOP_PER_POINT_EVAL
        f32_block = {"matrix_products": "v_mfma_f32_16x16x4_f32 only (csrc/gemm.hip, csrc/ode.hip)", "steps": k32,
                     "ms_per_step": round(1e3 * el32 / k32, 3), "value": round(world * B * k32 / el32, 3), "unit": "sequences/sec",
                     "cnf_launch_ms": round(cnf32, 3), "cnf_tflops": round(flop32 / (cnf32 * 1e-3) / 1e12, 3) if cnf32 > 0 else None,
                     "cnf_frac_of_f32_mfma_peak": round(flop32 / (cnf32 * 1e-3) / 1e12 / PEAK_MFMA_F32_TFLOPS, 4) if cnf32 > 0 else None}
        ops.set_matmul_mode(conv=prev[0], cnf=prev[1])

    # ---- the run-time accuracy guard (CaSPR.check_tol = 1e-5, ON by default since round 6: every solve repeated on 64 samples per frame at
    # half the steps on a side stream, compared on the device, verdict through the deferred channel as a RuntimeWarning).  The HEADLINE above
    # ran with it; here the same step with the guard OFF, for what it costs -- and what it reported during the headline's timed steps
    guard_block = None
    if not args.no_guard_subblock and model.check_tol is not None:
        import warnings
        with warnings.catch_warnings(record=True) as wrec:
            warnings.simplefilter("always")
            ops.check_deferred_errors()
        report, worst = {k: dict(v) for k, v in ops.GUARD_LAST.items()}, ops.GUARD_HISTORY_MAX
        spoke = [str(w.message)[:200] for w in wrec if "not converged" in str(w.message)] + guard_warnings
        tol_on, model.check_tol = model.check_tol, None
        step()
        kg = max(1, min(args.steps, 5))
        elg, _ = timed_steps(kg)
        model.check_tol = tol_on
        guard_block = {"check_tol": tol_on, "latent_check_tol": 100.0 * tol_on, "check_action": model.check_action, "check_points_per_frame": model.check_points,
                       "steps_guard_off": kg, "ms_per_step": round(1e3 * elapsed / args.steps, 3), "ms_per_step_guard_off": round(1e3 * elg / kg, 3),
                       "overhead_frac": round((elapsed / args.steps) / (elg / kg) - 1.0, 4), "verdict": "quiet" if not spoke else "warned: " + spoke[0],
                       "worst_estimate_over_bound": round(worst, 5), "report": report,
                       "what": "the headline's timed steps ran WITH the guard (the model's default); max |x_S - x_{S/2}| / 15 (Richardson estimate of the "
                               "delivered S-step solution) per solve, no host synchronisation in the path"}

    # ---- the other workloads BASELINE.json names, as sub-blocks of the same driver-run line (one GPU only; after the timed region)
    extra = {}
    default_workload = (args.clouds == "cars" and (B, T, N) == (16, 10, 2048) and not args.weights and not args.calibrate_cnf_steps)
    if world == 1 and not args.no_sub_blocks and default_workload:
        extra = sub_blocks(args, dev, ops, timed_steps, x, ts)

    # ---- N > 1: the one collective north_star names -- cfg-3's training step SHARDED over the ranks (8 sequences of T=10, N=1024 per rank,
    # forward + HIP backward + ONE flat 65 MB gradient all-reduce over RCCL + Adam; bench_train.py's measure(), every rank takes part).
    # The inference value above has no data-path collective (sequences are independent); this is where xGMI carries data.
    train_sharded = None
    if world > 1 and not args.no_train_subblock:
        import types
        import bench_train
        tb, tt, tn = (int(v) for v in args.train_shape.split(","))
        targs = types.SimpleNamespace(batch=tb, seq_len=tt, num_pts=tn, cnf_steps=8, latent_steps=2, mode="full", steps=3, warmup=1)
        torch.cuda.empty_cache()
        torch.cuda.reset_peak_memory_stats()
        tm = bench_train.measure(targs, dev, rank, world)
        tms, troof = bench_train.summarize(targs, tm, world)
        train_sharded = {"workload": "cfg-3 (BASELINE.json configs[2]) sharded: run_one_epoch body (train_utils.py:120-176), %d sequences/rank x %d ranks, T=%d, N=%d, "
                                     "forward + backward + ONE all-reduce of the flat gradient bucket + Adam; seeded random-init weights" % (tb, world, tt, tn),
                         "steps": targs.steps, "warmup": targs.warmup, "ms_per_step": round(tms, 3),
                         "value": None if args.share_gpu else round(world * tb * targs.steps / tm["elapsed"], 3), "unit": "sequences/sec",
                         "allreduce_ms": tm["allreduce_ms"], "allreduce_ms_min_over_ranks": min(tm["allreduce_ms"]), "bucket_bytes": tm["bucket_bytes"],
                         "allreduce_note": "per rank, mean over the steps, HIP events around the collective on the launch stream (gloo test mode: host "
                                           "wall clock); a rank that arrives early waits inside it, so the minimum over ranks is closest to the transport",
                         "ranks": {"ms_per_step": tm["per_rank_ms"], "library": ("gloo (--share-gpu test mode)" if args.share_gpu else tm["collective_library"])},
                         "loss_first": tm["losses"][0], "loss_last": tm["losses"][-1], "roofline": troof,
                         "max_mem_GB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1)}
        del tm
        torch.cuda.empty_cache()

    rc = 0
    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        value = world * B * args.steps / elapsed
        # ---- roofline of the dominant kernel (the CNF solve), HIP events recorded on the launch stream
        ev = timers.get("cnf_rk4", [])
        cnf_ms = sum(a.elapsed_time(b) for a, b in ev) / max(len(ev), 1)
        launches_per_step = max(len(ev) // max(args.steps, 1), 1)
        flop = float((hi - lo) * T * N) * 4 * args.cnf_steps * CNF_FLOP_PER_POINT_EVAL / launches_per_step
        achieved = flop / (cnf_ms * 1e-3) / 1e12 if cnf_ms > 0 else 0.0
        x6 = mode["cnf"] == "bf16x6"
        peak = PEAK_MFMA_BF16_TFLOPS / 6.0 if x6 else PEAK_MFMA_F32_TFLOPS
        # HBM traffic per launch cannot be counted from inside this process: it comes from the committed PMC passes of this same
        # command and workload (FETCH_SIZE / WRITE_SIZE in separate rocprofv3 passes, gfx950 correction), keyed by kernel + shape
        traffic, traffic_src = None, None
        tpath = os.path.join(ROOT, "profiles", "kernel_traffic.json")
        traffic_table = json.load(open(tpath)) if os.path.exists(tpath) else {}
        tj = traffic_table.get("%s:%dx%dx%d:s%d" % ("cnf_rk4_x6w_kernel" if x6 else "cnf_rk4_kernel", hi - lo, T, N, args.cnf_steps))
        if tj:
            traffic = int(1024 * (tj["fetch_size_kb_per_launch"] * tj["fetch_correction"] + tj["write_size_kb_per_launch"]))
            traffic_src = tj["source"]
        roofline = {"kernel": "cnf_rk4_x6w_kernel (csrc/ode_bf16x6w.hip)" if x6 else "cnf_rk4_kernel<false> (csrc/ode.hip)",
                    "bound": "mfma", "achieved": round(achieved, 3), "peak": round(peak, 1), "unit": "TFLOP/s", "frac": round(achieved / peak, 4),
                    "peak_note": ("dense bf16 MFMA peak 2500 TFLOP/s / 6 partial products per f32 product = 416.7 f32-equivalent TFLOP/s"
                                  if x6 else "dense f32-input MFMA peak"),
                    "traffic": traffic, "traffic_unit": "bytes/launch", "traffic_source": traffic_src,
                    "launch_ms": round(cnf_ms, 3), "launches_timed": len(ev), "flop_per_launch": flop}
        breakdown = {k: round(sum(a.elapsed_time(b) for a, b in v) / 2, 3) for k, v in stage_timers.items()}
        breakdown["cnf_rk4"] = round(sum(a.elapsed_time(b) for a, b in timers.get("cnf_rk4", [])) / args.steps, 3)   # the timed steps' own
        roofline["kernels"] = kernel_rooflines(roofline, detail, traffic_table, (hi - lo, T, N), breakdown.get("enc_set_abstraction"))

        cpu, parity_ok = None, None
        if extra.get("stress_dynamics") and not extra["stress_dynamics"]["parity"]["ok"]:
            rc = 1
        tc_ = extra.get("trained_checkpoint")
        if tc_ and not (tc_["parity_hip_vs_f64_oracle"]["ok"] and tc_["at_calibrated_steps"]["parity_hip_vs_f64_oracle"]["ok"]
                        and tc_["checkpoint_round_trip_bitwise"] and tc_["train_finite"]
                        and tc_["held_out_chamfer_x1000"]["after"] < tc_["held_out_chamfer_x1000"]["before"]
                        and tc_["at_calibrated_steps"]["guard"]["verdict"] == "quiet"):
            rc = 1            # the trained-checkpoint leg: parity on the trained weights at 8 / 2 AND at the calibrated counts, the round trip,
                              # a finite run whose held-out Chamfer improved (single-step losses on fresh batches are too noisy to gate on),
                              # and a guard that is quiet at the counts the calibration installed
        if extra.get("cfg5") and not extra["cfg5"]["parity"]["ok"]:
            rc = 1
        if guard_block is not None and guard_block["verdict"] != "quiet" and not args.weights and not args.calibrate_cnf_steps:
            rc = 1            # seeded weights at 8 / 2 steps are converged: a guard that speaks here is a bug
        if not args.no_cpu_baseline:
            cpu, parity_ok = cpu_baseline_and_parity(args, model, sd, ops, dev, out, x, x_all, sp_all, out[0], times_cpu, ts, T, N, dense_sequences)
            if not parity_ok:
                rc = 1

        cfg_name = "cars.cfg rigid recon (BASELINE.json configs[1])" if args.clouds == "cars" else "synthetic random clouds (BASELINE.json configs[4])"
        print(json.dumps({
            "metric": "sequences/sec (CaSPR.reconstruct, %s T=%d N=%d)" % ("rigid-cars" if args.clouds == "cars" else "random clouds", T, N),
            "value": None if args.share_gpu else round(value, 3), "unit": "sequences/sec",
            "test_mode": "ranks share one device over gloo (--share-gpu): NOT a measurement" if args.share_gpu else None,
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if not (x6 or mode["conv"] == "bf16x6") else
                     "f32 (operands split exactly into 3 bf16, 6 bf16-MFMA partial products per product, f32 accumulate)",
            "data": "synthetic",
            "config": {"workload": "%s: reconstruct(), B=%d sequences/GPU, T=%d, N=%d, num_points=%d, all steps observed; %s"
                                   % (cfg_name, B, T, N, N, weights_desc),
                       "global_batch": world * B, "seq_len": T, "num_pts": N, "cnf_rk4_steps": args.cnf_steps,
                       "latent_rk4_steps": args.latent_steps, "cnf_divergence": "skipped (sampling)", "parallelism": "seq-shard x%d" % world,
                       "ranks": {"ms_per_step": per_rank_ms, "collectives": "none in the data path; barrier + max-over-ranks timing only",
                                 "library": ("gloo (--share-gpu test mode)" if args.share_gpu else collective_library()) if world > 1 else None},
                       "base_samples": "drawn in-step (CPU generator, models/utils.py:25), pinned buffer + async copy under the encoder",
                       "matrix_products": dict(mode, selection=kernel_selection()), "calibration": calibration,
                       "box": box_calibration() if (world == 1 and not args.no_cpu_baseline) else None,
                       "nfe": [int(v) for v in model.get_nfe()]},
            "roofline": roofline, "f32_mfma_path": f32_block, "cfg5": extra.get("cfg5"), "train_cfg3": extra.get("train_cfg3") if world == 1 else train_sharded,
            "stress_dynamics": extra.get("stress_dynamics"), "trained_checkpoint": extra.get("trained_checkpoint"), "accuracy_guard": guard_block, "cpu_baseline": cpu, "parity_ok": parity_ok, "stage_ms_per_step": breakdown,
        }))
        sys.stdout.flush()
    ops.check_deferred_errors()
    if world > 1:
        flag = torch.tensor([rc], device=torch.device("cpu") if args.share_gpu else dev)
        dist.broadcast(flag, 0)
        rc = int(flag.item())
        dist.barrier()
        dist.destroy_process_group()
    sys.exit(rc)


def kernel_selection():
    """caspr_amd.config.active(): the kernel / schedule selection in force (one configuration object; the environment is read only
    under CASPR_DEBUG=1)."""
    from caspr_amd import config
    return config.active()


def sub_blocks(args, dev, ops, timed_steps, x_headline, ts_headline):
    """cfg5 / train_cfg3 / stress_dynamics: the other two throughput configurations of BASELINE.json and the stress-dynamics
    regime, measured by the SAME driver-run command as the headline (round-3 review: they existed only as builder-run files).
    Each is a few steps, bracketed like the headline's (synchronize on both sides), after the headline's timed region."""
    import types
    from caspr_amd.models import CaSPR
    from caspr_amd.utils.synthetic import seeded_state_dict, stress_state_dict, random_clouds
    out = {}
    peak_x6 = PEAK_MFMA_BF16_TFLOPS / 6.0

    tpath = os.path.join(ROOT, "profiles", "kernel_traffic.json")
    traffic_table = json.load(open(tpath)) if os.path.exists(tpath) else {}

    def cnf_entry(frames, n, steps, shape):
        ev = ops.TIMERS.get("cnf_rk4", [])
        ms = sum(a.elapsed_time(b) for a, b in ev) / max(len(ev), 1)
        flop = float(frames * n) * 4 * steps * CNF_FLOP_PER_POINT_EVAL
        ach = flop / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        # HBM-side bytes per launch from the committed PMC passes of this shape (FETCH_SIZE / WRITE_SIZE in separate rocprofv3 passes)
        tj = traffic_table.get("cnf_rk4_x6w_kernel:%dx%dx%d:s%d" % (shape + (steps,)))
        if tj is None:
            # the counter pass was taken at another step count of the same shape (the kernel's HBM-side bytes per launch -- state in / out, weights,
            # per-frame tables -- do not depend on it: 22.8 MB raw at S = 64 against 23.0 at S = 8): the nearest one, named as such
            pre = "cnf_rk4_x6w_kernel:%dx%dx%d:s" % shape
            near = sorted((abs(int(k[len(pre):]) - steps), k) for k in traffic_table if k.startswith(pre))
            if near:
                tj = dict(traffic_table[near[0][1]])
                tj["source"] = "%s (pass taken at S = %s; this launch ran S = %d)" % (tj["source"], near[0][1][len(pre):], steps)
        return {"kernel": "cnf_rk4_x6w_kernel", "bound": "mfma", "achieved": round(ach, 3), "peak": round(peak_x6, 1), "unit": "TFLOP/s",
                "frac": round(ach / peak_x6, 4), "launch_ms": round(ms, 3), "flop_per_launch": flop,
                "traffic": int(1024 * (tj["fetch_size_kb_per_launch"] * tj["fetch_correction"] + tj["write_size_kb_per_launch"])) if tj else None,
                "traffic_unit": "bytes/launch", "traffic_source": tj["source"] if tj else None}

    # ---- cfg5 (BASELINE.json configs[4]): one GPU's share of B=512 over 8 GPUs = 64 sequences, T=20, N=4096, i.i.d. random clouds
    B5, T5, N5 = 64, 20, 4096
    m5 = CaSPR(cnf_rk4_steps=args.cnf_steps, latent_rk4_steps=args.latent_steps)
    m5.load_state_dict(seeded_state_dict(m5.state_dict(), 0))
    m5 = m5.to(dev).eval()
    x5 = random_clouds(B5, T5, N5, seed=1234)
    ts5 = (x5[0, :, 0, 3] / 5.0).to(dev)
    x5 = x5.to(dev)

    def step5():
        with torch.no_grad():
            return m5.reconstruct(x5, num_points=N5, timestamps=ts5)
    step5()
    el, o5 = timed_steps(2, step5)
    roof5 = cnf_entry(B5 * T5, N5, args.cnf_steps, (B5, T5, N5))
    finite = bool(torch.isfinite(o5[2]).all()) and bool(torch.isfinite(o5[3]).all())
    # asserted in-run (round-4 review: finiteness only): (i) the LAST sequence of the batch re-run on its own gives its part of the
    # full-batch outputs bit for bit (sequences are independent: what holds for sequence 63 of 64 holds for any shard of the 512);
    # (ii) the first and the last sequence against the f64 evaluation of the reference graph, flat 1e-5, on 64 of their samples
    from oracle import model as O5
    with torch.no_grad():
        alone = m5.reconstruct(x5[B5 - 1:], num_points=N5, timestamps=ts5, y=o5[0][B5 - 1:])
        shard_ok = bool(torch.equal(alone[2], o5[2][B5 - 1:])) and bool(torch.equal(alone[3], o5[3][B5 - 1:]))
        pick = [0, B5 - 1]
        yb5 = o5[0][pick][:, :, :64].contiguous()
        _, _, gx5, gt5 = m5.reconstruct(x5[pick], num_points=64, timestamps=ts5, y=yb5)
    sd64_5 = {k: v.detach().cpu().double() for k, v in m5.state_dict().items()}
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    _, _, x64_5, t64_5 = O5.reconstruct(sd64_5, x5[pick].cpu().double(), yb5.cpu().double(), timestamps=ts5.cpu().double(), cnf_steps=args.cnf_steps,
                                        latent_steps=args.latent_steps)
    ex5, et5 = float((gx5.cpu().double() - x64_5).abs().max()), float((gt5.cpu().double() - t64_5).abs().max())
    ok5 = finite and shard_ok and ex5 <= 1e-5 and et5 <= 1e-5
    out["cfg5"] = {"workload": "synthetic random clouds (BASELINE.json configs[4]), one GPU's share: reconstruct(), B=%d, T=%d, N=%d, num_points=%d, "
                               "seeded random-init weights" % (B5, T5, N5, N5), "steps": 2, "warmup": 1, "ms_per_step": round(1e3 * el / 2, 3),
                   "value": round(B5 * 2 / el, 3), "unit": "sequences/sec", "roofline": roof5, "outputs_finite": finite,
                   "parity": {"sequences_checked": pick, "samples_per_frame": 64, "x_hip_vs_f64": ex5, "tnocs_hip_vs_f64": et5, "bound": 1e-5,
                              "last_sequence_alone_bitwise": shard_ok, "ok": bool(ok5),
                              "also": "tests/test_hip_parity.py::test_cfg5_random_clouds (2 x 20 x 4096: indices bit-exact, flat 1e-5, sharding invariance)"},
                   "max_mem_GB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1)}
    del m5, x5, o5, alone
    torch.cuda.empty_cache()

    # ---- train_cfg3 (configs[2]): one rank's shard of the B=64 training step, (8, 10, 1024): forward + HIP backward + Adam
    import bench_train
    targs = types.SimpleNamespace(batch=8, seq_len=10, num_pts=1024, cnf_steps=8, latent_steps=2, mode="full", steps=3, warmup=1)
    torch.cuda.reset_peak_memory_stats()
    tm = bench_train.measure(targs, dev, 0, 1)
    tms, troof = bench_train.summarize(targs, tm, 1)
    out["train_cfg3"] = {"workload": "cfg-3 shard (BASELINE.json configs[2]): run_one_epoch body (train_utils.py:120-176), B=8 sequences/GPU, T=10, N=1024, NLL (CNF "
                                     "with Hutchinson divergence) + T-NOCS L1, forward + backward + Adam; seeded random-init weights",
                         "steps": targs.steps, "warmup": targs.warmup, "ms_per_step": round(tms, 3), "value": round(8 * targs.steps / tm["elapsed"], 3),
                         "unit": "sequences/sec", "roofline": troof, "loss_first": tm["losses"][0], "loss_last": tm["losses"][-1],
                         "max_mem_GB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1)}
    first_loss = tm["losses"][0]
    del tm
    torch.cuda.empty_cache()
    # the same step with the CNF's tape kept per RK4 step and recomputed in the backward pass (config.train_cnf_checkpoint; the
    # reference's adjoint is O(1) memory too, cnf.py:100): what the memory costs in time
    from caspr_amd.train import flow_grad
    targs2 = types.SimpleNamespace(batch=8, seq_len=10, num_pts=1024, cnf_steps=8, latent_steps=2, mode="full", steps=2, warmup=1)
    prev_ck, flow_grad.CHECKPOINT_STEPS = flow_grad.CHECKPOINT_STEPS, True
    try:
        torch.cuda.reset_peak_memory_stats()
        tm2 = bench_train.measure(targs2, dev, 0, 1)
        out["train_cfg3"]["checkpointed_steps"] = {"ms_per_step": round(1e3 * tm2["elapsed"] / targs2.steps, 3),
                                                   "max_mem_GB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1),
                                                   "loss_first_identical": bool(tm2["losses"][0] == first_loss),
                                                   "what": "caspr_amd.config.train_cnf_checkpoint = True: state per RK4 step only, the step's four evaluations recomputed in the backward pass"}
        del tm2
    finally:
        flow_grad.CHECKPOINT_STEPS = prev_ck
    torch.cuda.empty_cache()

    # ---- stress dynamics: the headline workload on weights whose flow is HARD to integrate (synthetic.stress_state_dict); the step
    # count is chosen by step doubling at 1e-5 (CaSPR.calibrate_rk4_steps), the throughput quoted at THAT count next to the fixed 8
    from oracle import model as O
    ms_ = CaSPR(cnf_rk4_steps=args.cnf_steps, latent_rk4_steps=args.latent_steps)
    ssd = stress_state_dict(ms_.state_dict(), 0)
    ms_.load_state_dict(ssd)
    ms_ = ms_.to(dev).eval()
    torch.manual_seed(4)
    S, diffs, L, ldiffs = ms_.calibrate_rk4_steps(x_headline, tol=1e-5, timestamps=ts_headline, latent_tol=1e-4)
    Bh, Th, Nh = x_headline.shape[:3]

    def step_s():
        with torch.no_grad():
            return ms_.reconstruct(x_headline, num_points=Nh, timestamps=ts_headline)
    step_s()
    el, os_ = timed_steps(2, step_s)
    roof = cnf_entry(Bh * Th, Nh, S, (Bh, Th, Nh))
    # parity of this regime on sequence 0, 64 samples per frame: against the f64 oracle at the SAME step counts (flat 1e-5), and the CNF
    # on the HIP path's own latent codes against the converged f64 solution (256 steps) and the oracle's dopri5(1e-5)
    yb = os_[0][:1, :, :64].contiguous()
    with torch.no_grad():
        _, _, gx, gt = ms_.reconstruct(x_headline[:1], num_points=64, timestamps=ts_headline, y=yb)
        z0, _ = ms_.encode(x_headline[:1])
        z = ms_.aggregate_and_solve_latent(z0, ts_headline.view(1, -1))
    sd64 = {k: v.double() for k, v in ssd.items()}
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    _, _, x64, t64 = O.reconstruct(sd64, x_headline[:1].cpu().double(), yb.cpu().double(), timestamps=ts_headline.cpu().double(), cnf_steps=S, latent_steps=L)
    ctx = z.cpu().double().view(Th, -1)
    yy = yb.cpu().double().view(Th, 64, 3)
    cnt = [0]
    dop = O.point_cnf(sd64, yy, ctx, None, True, "dopri5", counter=cnt)
    conv = O.point_cnf(sd64, yy, ctx, None, True, "rk4", 256)
    g = gx.cpu().double().view(Th, 64, 3)
    ex, et = float((gx.cpu().double() - x64).abs().max()), float((gt.cpu().double() - t64).abs().max())
    e_conv, e_dop = float((g - conv).abs().max()), float((dop - conv).abs().max())
    ok = ex <= 1e-5 and et <= 1e-5 and e_conv <= 1e-5 + 2.0 * diffs[S] and S > 8
    # the run-time guard on this regime: at the headline's fixed S = 8 it must speak (the true error is 6e-3), at the calibrated S it is quiet
    guard = {}
    ms_.check_action = "raise"
    for name, (Sg, Lg) in (("fixed_8_steps", (args.cnf_steps, L)), ("calibrated", (S, L))):
        ops.reset_guard()
        for b_ in ms_.point_cnf.chain:
            if hasattr(b_, "rk4_steps"):
                b_.rk4_steps = Sg
        ms_.latent_ode.rk4_steps = Lg
        with torch.no_grad():
            ms_.reconstruct(x_headline[:2], num_points=256, timestamps=ts_headline, check_tol=1e-5)
        try:
            ops.check_deferred_errors()
            verdict = "quiet"
        except ops.CasprAccuracyError:
            verdict = "raised"
        guard[name] = {"cnf_rk4_steps": Sg, "verdict": verdict, "cnf_estimate": ops.GUARD_LAST.get("cnf", {}).get("estimate"),
                       "latent_estimate": ops.GUARD_LAST.get("latent", {}).get("estimate")}
    ok = ok and guard["fixed_8_steps"]["verdict"] == "raised" and guard["calibrated"]["verdict"] == "quiet"
    out["stress_dynamics"] = {
        "accuracy_guard": guard,
        "workload": "the headline workload (B=%d, T=%d, N=%d) on the STRESS weights (caspr_amd.utils.synthetic.stress_state_dict: time-switching gates, "
                    "saturated softplus tails, T_end = 1, a latent field that moves)" % (Bh, Th, Nh),
        "calibration": {"tol": 1e-5, "cnf_rk4_steps": S, "step_doubling_diffs": {str(k): v for k, v in diffs.items()}, "latent_tol": 1e-4,
                        "latent_rk4_steps": L, "latent_step_doubling_diffs": {str(k): v for k, v in ldiffs.items()},
                        "reference_dopri5_nfe": cnt[0], "rk4_nfe": 4 * S},
        "steps": 2, "warmup": 1, "ms_per_step": round(1e3 * el / 2, 3), "value": round(Bh * 2 / el, 3), "unit": "sequences/sec", "roofline": roof,
        "parity": {"sequence": 0, "samples_per_frame": 64, "x_hip_vs_f64_same_steps": ex, "tnocs_hip_vs_f64": et, "bound": 1e-5,
                   "cnf_hip_vs_converged_f64_rk4_256": e_conv, "oracle_dopri5_1e-5_vs_converged": e_dop,
                   "cnf_hip_vs_oracle_dopri5": float((g - dop).abs().max()), "ok": bool(ok)}}
    del ms_
    torch.cuda.empty_cache()

    # ---- a checkpoint that went through the whole surface (the pretrained caspr_weights_cars.pth cannot be fetched offline): 2000 training
    # steps on fresh synthetic car sequences with the HIP training tier -- long enough for the flow to become HARD to integrate: the default
    # 8 steps no longer meet 1e-5 there -- written in the reference's format, loaded back as `--weights` loads one, evaluated held out, the step
    # counts chosen by the refined step-doubling calibration, and THE HEADLINE CALL TIMED AT THOSE COUNTS by this file's own bracketed clock
    # (tools/train_and_eval.py)
    import importlib.util
    import tempfile
    spec = importlib.util.spec_from_file_location("train_and_eval", os.path.join(ROOT, "tools", "train_and_eval.py"))
    tae = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tae)
    with tempfile.TemporaryDirectory() as td:
        import contextlib
        import io
        with contextlib.redirect_stdout(io.StringIO()):
            probe_at = ",".join(str(v) for v in (1200, 1600) if v < args.trained_steps)
            rep = tae.main(["--steps", str(args.trained_steps), "--eval-seqs", "2", "--ckpt", os.path.join(td, "time_model_0.pth"), "--no-dopri5",
                            "--probe-at", probe_at], timed_steps=timed_steps)
    curve = rep["train"]["curve"]
    head = rep["headline_on_trained_weights"]
    cal = dict(head["at_calibrated_steps"])
    cal["parity_hip_vs_f64_oracle"] = rep["parity_trained_weights"]["hip_vs_f64_oracle_at_calibrated_steps"]
    cal["cnf_evaluations"] = 4 * cal["cnf_rk4_steps"]
    out["trained_checkpoint"] = {
        "what": "%d training steps (B=8, T=10, N=1024, fresh synthetic cars per step, train_utils.py's loss, Adam 1e-4) -> reference-format checkpoint "
                "-> load -> held-out evaluation at 10 x 2048 (evaluations.py protocol) -> calibrate_rk4_steps(tol=1e-5, refined between the powers "
                "of two) -> the headline call (B=16, T=10, N=2048, guard on) at the default and at the calibrated counts; tools/train_and_eval.py"
                % args.trained_steps,
        "train_steps": args.trained_steps, "train_wall_s": rep["train"]["wall_s"], "train_finite": rep["train"]["finite"],
        "loss_first": curve[0]["loss"], "loss_last": curve[-1]["loss"],
        "loss_mean_first_5_records": sum(c["loss"] for c in curve[:5]) / len(curve[:5]), "loss_mean_last_5_records": sum(c["loss"] for c in curve[-5:]) / len(curve[-5:]),
        "checkpoint_round_trip_bitwise": rep["checkpoint"]["round_trip_bitwise"], "checkpoint_keys": rep["checkpoint"]["keys"],
        "held_out_chamfer_x1000": {"before": rep["held_out_before"]["chamfer_x1000"]["mean"], "after": rep["held_out_after"]["chamfer_x1000"]["mean"]},
        "held_out_tnocs_l2": {"before": rep["held_out_before"]["tnocs_space_l2"]["mean"], "after": rep["held_out_after"]["tnocs_space_l2"]["mean"]},
        "calibration_tol_1e-5": rep["calibration_tol_1e-5"], "guard_at_8_and_2_steps": rep["guard_at_8_and_2_steps"],
        "parity_hip_vs_f64_oracle": rep["parity_trained_weights"]["hip_vs_f64_oracle_same_rk4_map"],
        "at_default_steps": {k: head[k] for k in ("cnf_rk4_steps", "latent_rk4_steps", "steps", "ms_per_step", "sequences_per_sec", "guard")},
        "at_calibrated_steps": cal,
        # the SAME training run calibrated on the way (the run is not disturbed: counts and generator states restored): the flow gets harder to
        # integrate as it trains, and not smoothly -- the calibrated count, and with it the throughput at 1e-5, is a property of the checkpoint
        "along_training": rep["along_training"] + [{"train_steps": args.trained_steps, "cnf_rk4_steps": cal["cnf_rk4_steps"], "latent_rk4_steps": cal["latent_rk4_steps"],
                                                    "ms_per_step": cal["ms_per_step"], "sequences_per_sec": cal["sequences_per_sec"]}]}
    torch.cuda.empty_cache()
    return out


def kernel_rooflines(cnf, detail, traffic_table, shape, sa_wall_ms=None):
    """achieved / peak / frac for the matrix kernels behind the CNF solve, from the detail pass' per-launch HIP events:
    the largest pointwise conv on the bf16x6 kernel (the 1600 -> 1600 head layer at cfg-2) and all of them together
    (2 Cin Cout FLOP per row; conv -> GroupNorm calls include their statistics epilogue and finalize kernel), and the fused
    set-abstraction kernels (f32 MFMA; 2 FLOP per multiply-add over every gathered sample).  `traffic` = HBM-side bytes per
    launch from the committed PMC passes (profiles/kernel_traffic.json, keyed by kernel + workload shape), or null."""
    out = [{k: cnf[k] for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "launch_ms")}]
    wl = "%dx%dx%d" % shape
    convs = {}
    for k, ms in detail.items():
        p = k.split(":")
        if p[1] == "conv1x1_bf16x6":
            convs.setdefault((int(p[2]), int(p[3]), int(p[4])), []).extend(ms)
    if convs:
        flop_all = sum(2.0 * ci * co * rows * len(ms) for (ci, co, rows), ms in convs.items())
        ms_all = sum(sum(ms) for ms in convs.values())
        (ci, co, rows), ms = max(convs.items(), key=lambda kv: kv[0][0] * kv[0][1] * kv[0][2])
        a_big = 2.0 * ci * co * rows / (sum(ms) / len(ms) * 1e-3) / 1e12
        # per LAYER (round 6): the sum over the launches of one conv -> GroupNorm call of that shape, run on its own between marker launches
        # (tools/conv_layers_pmc.py; rounds 3-5 averaged every launch of the persistent kernel, which hid the pieces that re-read their input)
        def layer_traffic(ci_, co_, rows_):
            t_ = traffic_table.get("conv_layer:%d:%d:%d:%s" % (ci_, co_, rows_, wl))
            return None if not t_ else {"bytes": int(1024 * (t_["fetch_size_kb"] * t_["fetch_correction"] + t_["write_size_kb"])),
                                        "over_algorithmic": round((t_["fetch_size_kb"] * t_["fetch_correction"] + t_["write_size_kb"]) / t_["algorithmic_kb"], 3)}
        big_t = layer_traffic(ci, co, rows)
        tj = None if big_t else traffic_table.get("conv_largest:%dx%d:%s" % (ci, co, wl))
        out.append({"kernel": "pointwise convs on the bf16x6 kernels (csrc/gemm_bf16x6w.hip for >= 1024 input and >= 512 output channels, csrc/gemm_bf16x6.hip "
                              "otherwise); largest layer %d -> %d over %d rows" % (ci, co, rows), "bound": "mfma",
                    "achieved": round(a_big, 3), "peak": round(PEAK_MFMA_BF16_TFLOPS / 6.0, 1), "unit": "TFLOP/s", "frac": round(a_big / (PEAK_MFMA_BF16_TFLOPS / 6.0), 4),
                    "traffic": big_t["bytes"] if big_t else (int(1024 * (tj["fetch_size_kb_per_launch"] * tj["fetch_correction"] + tj["write_size_kb_per_launch"])) if tj else None),
                    "traffic_over_algorithmic": big_t["over_algorithmic"] if big_t else None,
                    "traffic_unit": "bytes per conv -> GroupNorm call of the largest layer, all of its launches (main tiles + channel remainder + finalize)",
                    "launch_ms": round(sum(ms) / len(ms), 3),
                    "all_layers": {"launches_per_step": sum(len(m) for m in convs.values()) // 2, "ms_per_step": round(ms_all / 2, 3),
                                   "achieved": round(flop_all / (ms_all * 1e-3) / 1e12, 3), "frac": round(flop_all / (ms_all * 1e-3) / 1e12 / (PEAK_MFMA_BF16_TFLOPS / 6.0), 4)},
                    # per layer shape (Cin -> Cout over rows): launches per step, ms per launch, fraction of the bf16x6 ceiling
                    "layers": [{"cin": ci_, "cout": co_, "rows": rows_, "launches_per_step": len(m_) // 2, "ms": round(sum(m_) / len(m_), 4),
                                "frac": round(2.0 * ci_ * co_ * rows_ / (sum(m_) / len(m_) * 1e-3) / 1e12 / (PEAK_MFMA_BF16_TFLOPS / 6.0), 4),
                                "traffic": (layer_traffic(ci_, co_, rows_) or {}).get("bytes"),
                                "traffic_over_algorithmic": (layer_traffic(ci_, co_, rows_) or {}).get("over_algorithmic")}
                               for (ci_, co_, rows_), m_ in sorted(convs.items(), key=lambda kv: -kv[0][0] * kv[0][1] * kv[0][2])]})
    sa = [(float(k.split(":")[5]) * 1e6, ms) for k, ms in detail.items() if k.split(":")[1].startswith("sa_mlp_max")]     # (+ "_mfma" / "_f64": a scale's two halves on two streams)
    if sa:
        flop = sum(f * len(ms) for f, ms in sa)
        ms_all = sum(sum(ms) for _, ms in sa)
        a_sa = flop / (ms_all * 1e-3) / 1e12
        tj = traffic_table.get("sa_all:%s" % wl)
        out.append({"kernel": "sa_small_kernel / sa_mlp_kernel (csrc/sa_mlp.hip), all fused set-abstraction launches", "bound": "mfma", "achieved": round(a_sa, 3),
                    "peak": PEAK_MFMA_F32_TFLOPS, "unit": "TFLOP/s", "frac": round(a_sa / PEAK_MFMA_F32_TFLOPS, 4),
                    "traffic": int(1024 * (tj["fetch_size_kb_per_step"] * tj["fetch_correction"] + tj["write_size_kb_per_step"])) if tj else None,
                    "traffic_unit": "bytes/step (all launches)",
                    "launches_per_step": sum(len(ms) for _, ms in sa) // 2,
                    # the two scales of a level run on two streams since round 5: the sum of the launches' own durations counts the
                    # overlap twice; the wall time of the five levels on the main stream is what the step pays
                    "ms_per_step": round(sa_wall_ms, 3) if sa_wall_ms else round(ms_all / 2, 3), "sum_of_launch_ms_per_step": round(ms_all / 2, 3),
                    "achieved_over_wall": round(flop / 2 / (sa_wall_ms * 1e-3) / 1e12, 3) if sa_wall_ms else None,
                    "includes": "the f64 re-evaluation of balls with 2..8 distinct samples (sa_repair_f64_kernel) behind the register kernels"})
    return out


def box_calibration():
    """Bare v_mfma_f32_32x32x16_bf16 rate of THIS box on the operand mix of the bf16x6 scheme (tools/micro/mfma_power, ~100 ms per
    variant, built by __graft_entry__.build()): the boxes of the pool differ by 3-5 % in what their matrix pipes sustain under the
    power cap, and the headline moves with it.  None if the binary is not there."""
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools", "micro", "mfma_power")
    if not os.path.exists(exe):
        return None
    try:
        out = subprocess.run([exe], capture_output=True, text=True, timeout=60).stdout
        rate = {}
        for line in out.splitlines():
            p = line.split()
            if len(p) >= 6 and p[0] == "data":
                rate[p[1].rstrip(":")] = max(rate.get(p[1].rstrip(":"), 0.0), float(p[4]))
        if "7" not in rate:
            return None
        return {"bare_mfma_bf16_tflops": {"zeros": rate.get("0"), "random_sign_exponent_mantissa": rate.get("3"), "bf16x6_operand_planes": rate.get("7")},
                "note": "register-only MFMA loop, one wave per SIMD, ~100 ms per variant (tools/micro/mfma_power.hip); the nominal peak is 2500"}
    except Exception:
        return None


def cpu_baseline_and_parity(args, model, sd, ops, dev, out, x, x_all, sp_all, ybase, times_cpu, ts, T, N, dense_sequences):
    """The CPU oracle on a bounded sample of the same workload (timed), and the asserted HIP-vs-oracle parity:
      * the FIRST and the LAST sequence of this rank's batch against the f64 evaluation of the reference graph (the oracle in
        double precision): |hip - f64| <= 1e-5 flat -- north_star's tolerance as written -- on sampled xyz, T-NOCS and
        Chamfer-L2.  The f32 oracle's own distance from f64 is printed next to it (on these sparse car clouds the reference's
        f32 arithmetic is 6e-5 / 3e-4 away: duplicate-padded neighbourhoods amplify f32 rounding inside GroupNorm);
      * every sequence of the full batch: bitwise equal to reconstructing the second half of the batch on its own
        (sequences are independent, so the two oracle-checked sequences stand for all B);
      * a well-conditioned (dense) cloud of the same shape: |hip - f64| <= DENSE_TOL / 2 on xyz and T-NOCS (the direct difference
        against the f32 oracle is reported beside it)."""
    from oracle import model as O
    DENSE_TOL = 1e-5      # north_star tolerance as written (the well-conditioned input; direct difference against the f32 oracle)
    cpu_model, total_cores = cpu_description()
    nb = x_all.shape[0]
    pick = [0, nb - 1] if nb > 1 else [0]
    xs, ys = x_all[pick], ybase[pick].cpu()
    # 16 intra-op threads: on the GPU box's host (profiles/r03_cpu_threads_probe.txt) the oracle does 0.18 / 0.21 / 0.17 / 0.12 / 0.04 /
    # 0.005 sequences/s at 8 / 16 / 32 / 64 / 128 / 256 threads on one sequence (round 3 timed 16 AND 32 and kept the better; they
    # trade places within 10 %, and the second trial cost 13 s of the driver's run)
    cpu_s, ncores = None, None
    for nt in sorted({min(total_cores, 16)}):
        torch.set_num_threads(nt)
        t1 = time.perf_counter()
        _, _, wx, wt = O.reconstruct(sd, xs, ys, timestamps=times_cpu, cnf_steps=args.cnf_steps, latent_steps=args.latent_steps)
        el_ = time.perf_counter() - t1
        if cpu_s is None or el_ < cpu_s:
            cpu_s, ncores = el_, nt
    torch.set_num_threads(min(total_cores, 32))
    gx, gt = out[2][pick].cpu(), out[3][pick].cpu()
    sd64 = {k: v.double() for k, v in sd.items()}
    _, _, x64, t64 = O.reconstruct(sd64, xs.double(), ys.double(), timestamps=times_cpu.double(), cnf_steps=args.cnf_steps,
                                   latent_steps=args.latent_steps)
    checks = []
    TOL = 1e-5            # north_star: "T-NOCS / CNF-sampled xyz within 1e-5 abs", asserted flat against the f64 evaluation
    # (configs[4], i.i.d. uniform clouds -- most neighbourhoods hold one to four points, every level's GroupNorm is degenerate, the f32
    # oracle is 6e-4 / 4e-3 from f64 -- carried a capped slack on T-NOCS until round 4 (1.5e-4 at B = 64); since round 5 it is flat like
    # everything else: 7e-7 measured)
    CAP = {}

    def cond(name, g, w32, w64):
        e_gpu, e_ref = float((g.double() - w64).abs().max()), float((w32.double() - w64).abs().max())
        bound = max(TOL, min(CAP[name], TOL + e_ref)) if name in CAP else TOL
        ok = e_gpu <= bound
        checks.append(ok)
        return {"hip_vs_f64": e_gpu, "oracle32_vs_f64": e_ref, "hip_vs_oracle32": float((g - w32).abs().max()), "bound": bound, "ok": ok}

    parity = {"sequences_checked": pick, "x": cond("x", gx, wx, x64), "tnocs": cond("tnocs", gt, wt, t64),
              "x_max_abs_err_vs_oracle32": float((gx - wx).abs().max()), "tnocs_max_abs_err_vs_oracle32": float((gt - wt).abs().max())}
    if sp_all is not None:   # Chamfer-L2 against the ground-truth NOCS points (evaluations.py:40-43)
        n = len(pick)
        gt_pts = sp_all[pick][:, :, :, :3].reshape(n * T, N, 3).contiguous()
        cd32 = O.chamfer_l2(wx.reshape(n * T, N, 3), gt_pts)
        cd64 = O.chamfer_l2(x64.reshape(n * T, N, 3).float(), gt_pts)
        d1, d2 = ops.chamfer_distance(out[2][pick].reshape(n * T, N, 3).contiguous(), gt_pts.to(dev))
        cd_gpu = (d1.mean(dim=1) + d2.mean(dim=1)).cpu()
        e_gpu, e_ref = float((cd_gpu - cd64).abs().max()), float((cd32 - cd64).abs().max())
        ok = e_gpu <= TOL
        checks.append(ok)
        parity["chamfer_l2"] = {"mean": float(cd_gpu.mean()), "hip_vs_f64": e_gpu, "oracle32_vs_f64": e_ref,
                                "hip_vs_oracle32": float((cd_gpu - cd32).abs().max()), "bound": TOL, "ok": ok}
    # full batch: the second half on its own must reproduce its part of the full-batch outputs bit for bit
    if nb >= 2:
        h = nb // 2
        with torch.no_grad():
            o2 = model.reconstruct(x[h:], num_points=N, timestamps=ts, y=ybase[h:])
        same = bool(torch.equal(o2[2], out[2][h:])) and (out[3] is None or bool(torch.equal(o2[3], out[3][h:])))
        checks.append(same)
        parity["full_batch_shard_invariance_bitwise"] = same
    # well-conditioned input of the same shape: direct bound
    xd, spd = dense_sequences(1, T, N, seed=4321)
    yd = ybase[:1].cpu()
    with torch.no_grad():
        od = model.reconstruct(xd.to(dev), num_points=N, timestamps=spd[0, :, 0, 3].to(dev), y=yd.to(dev))
    _, _, wxd, wtd = O.reconstruct(sd, xd, yd, timestamps=spd[0, :, 0, 3], cnf_steps=args.cnf_steps, latent_steps=args.latent_steps)
    _, _, xd64, td64 = O.reconstruct(sd64, xd.double(), yd.double(), timestamps=spd[0, :, 0, 3].double(), cnf_steps=args.cnf_steps,
                                     latent_steps=args.latent_steps)
    ex, et = float((od[2].cpu() - wxd).abs().max()), float((od[3].cpu() - wtd).abs().max())
    ex64, et64 = float((od[2].cpu().double() - xd64).abs().max()), float((od[3].cpu().double() - td64).abs().max())
    ox64, ot64 = float((wxd.double() - xd64).abs().max()), float((wtd.double() - td64).abs().max())
    # asserted against the f64 evaluation only, at HALF the tolerance; the direct HIP-vs-f32-oracle difference is reported (once the HIP
    # path is within a few 1e-6 of f64 it IS the f32 oracle's own error, which depends on the host's GEMM blocking and thread count)
    dense_ok = ex64 <= 0.5 * DENSE_TOL and et64 <= 0.5 * DENSE_TOL
    checks.append(dense_ok)
    parity["dense_input"] = {"x_max_abs_err": ex, "tnocs_max_abs_err": et, "criterion": DENSE_TOL, "x_hip_vs_f64": ex64, "tnocs_hip_vs_f64": et64,
                             "x_oracle32_vs_f64": ox64, "tnocs_oracle32_vs_f64": ot64, "ok": dense_ok}
    # NFE of the reference's adaptive solvers on this input (the oracle's dopri5 restatement, PARITY UNPINNED), small sample
    nfe = [0, 0]
    O.reconstruct(sd, xs[:1], ys[:1, :, :256].contiguous(), timestamps=times_cpu, method="dopri5", nfe=nfe)
    cpu = {"value": round(len(pick) / cpu_s, 5), "unit": "sequences/sec", "cores": ncores, "kind": "port",
           "host": {"cpu_model": cpu_model, "total_cores": total_cores, "threads_used": ncores},
           "sample": "%d sequences (T=%d, N=%d, num_points=%d) of the same workload through oracle.model.reconstruct "
                     "(torch-CPU + C point ops, same RK4 steps), %.1f s at %d intra-op threads" % (len(pick), T, N, N, cpu_s, ncores),
           "reference_dopri5_nfe": {"latent_ode": int(nfe[0]), "point_cnf": int(nfe[1]),
                                    "note": "function evaluations the reference's dopri5 (latent rtol=atol=1e-3, CNF 1e-5) spends on sequence 0 "
                                            "with 256 samples, oracle restatement; this build: config.nfe"},
           "parity": parity}
    return cpu, all(checks)


if __name__ == "__main__":
    main()
