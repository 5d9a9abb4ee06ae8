#!/usr/bin/env python
"""bench_train.py -- sequences/sec of one TRAINING step (BASELINE.json configs[2] = cfg-3).

    python bench_train.py --gpus N --steps K --warmup W [--mode full|pretrain]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench_train.py --gpus N --steps K --warmup W

Companion of bench.py (which measures the headline `reconstruct` metric and is the one the driver runs); same JSON
contract.  One "step" = `run_one_epoch`'s body (train_utils.py:120-176) on one batch resident in HBM: forward in
train() mode, loss = 0.01*mean(sum_n nll) + 100*mean(L1 tnocs), backward (HIP gradient kernels), gradient all-reduce
when N > 1 (ONE flat 65 MB bucket over RCCL -- the only collective), Adam.  Workload: cfg-3's per-GPU shard, 8 sequences
of T=10 x N=1024 per GPU (weak scaling), f32, 8 CNF RK4 steps with the Hutchinson divergence, 2 latent RK4 steps.

  roofline     : all matrix products of the step (forward + data gradient + weight gradient = 3x the forward
                 contraction FLOPs of SURVEY.md 8d: 176.6 GFLOP/sequence encoder, 21.70 GFLOP/sequence/evaluation CNF
                 with divergence) divided by the step time, vs the dense f32 MFMA peak;
  cpu_baseline : the CPU oracle's differentiable mode (oracle.model.training_loss + backward: torch-CPU autograd, a
                 port -- the reference's own training needs CUDA-only Kaolin ops) on ONE sequence of the same shape.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_MFMA_F32_TFLOPS = 157.3
ENC_GFLOP_PER_SEQ = {(10, 1024): 176.6, (10, 2048): 306.0, (20, 4096): 1129.6, (5, 512): 56.0}
CNF_FLOP_PER_POINT_EVAL_DIV = 2 * 2 * (3 * 512 + 512 * 512 + 512 * 512 + 512 * 3)               # value + tangent


def measure(args, dev, rank=0, world=1):
    """The timed training steps + the per-pipe detail pass; also called by bench.py for its `train_cfg3` sub-block (world = 1).
    -> dict with elapsed (max over ranks), per_rank_ms, losses, pipes, and what the CPU baseline needs."""
    from caspr_amd.models import CaSPR
    from caspr_amd.train.loop import GradBucket, train_step
    from caspr_amd.utils.sharding import max_over_ranks, per_rank_values, collective_library
    from caspr_amd.utils.synthetic import car_sequences, seeded_state_dict

    B, T, N = args.batch, args.seq_len, args.num_pts
    full = args.mode == "full"
    sd = seeded_state_dict(CaSPR().state_dict(), 0)
    model = CaSPR(pretrain_tnocs=not full, cnf_rk4_steps=args.cnf_steps, latent_rk4_steps=args.latent_steps)
    model.load_state_dict(sd if full else {k: v for k, v in sd.items() if k.startswith("encoder.")})
    model = model.to(dev).train()
    from caspr_amd.train.loop import broadcast_model
    broadcast_model(model, 0)       # replicas start from rank 0's parameters and buffers (train.py:131-132 replicates module 0)
    opt = torch.optim.Adam(model.parameters(), lr=1e-4, betas=(0.9, 0.999), eps=1e-8)
    bucket = GradBucket(model.parameters()) if world > 1 else None
    x_all, sp_all = car_sequences(world * B, T, N, seed=1234)
    x, sp = x_all[rank * B:(rank + 1) * B].to(dev), sp_all[rank * B:(rank + 1) * B].to(dev)
    e = torch.randn(B * T, N, 3, device=dev, generator=torch.Generator(device=dev).manual_seed(4321 + rank)) if full else None   # fixed Hutchinson noise: repeatable losses

    losses = []
    for _ in range(args.warmup):
        train_step(model, opt, x, sp, bucket=bucket, e=e)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
        bucket.timers = []             # the duration of every step's ONE collective (HIP events on the launch stream under RCCL)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        losses.append(train_step(model, opt, x, sp, bucket=bucket, e=e)[0])
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    el_local = time.perf_counter() - t0
    allreduce_ms, bucket_bytes = None, None
    if world > 1:
        from caspr_amd.train.loop import collective_ms
        ms_ = collective_ms(bucket.timers)
        bucket.timers = None
        # per rank: the mean over the steps (a rank that arrives early waits inside the collective for the slowest one: its figure holds
        # that wait; the MINIMUM over ranks is the closest to the transport's own time)
        allreduce_ms = [round(v, 3) for v in per_rank_values(sum(ms_) / max(len(ms_), 1), dev)]
        bucket_bytes = int(bucket.flat.numel() * 4)
    per_rank_ms = [round(1e3 * v / args.steps, 3) for v in per_rank_values(el_local, dev)]
    elapsed = max_over_ranks(el_local, dev)
    # ---- detail pass (not part of the timing above): one HIP-event pair per launch of every matrix kernel, by pipe
    from caspr_amd import ops
    torch.cuda.synchronize()
    ops.TIMERS.clear()
    ops.TIMING = 2
    train_step(model, opt, x, sp, bucket=bucket, e=e)
    torch.cuda.synchronize()
    ops.TIMING = False
    pipes = {"bf16x6": [0.0, 0.0, 0], "f32_mfma": [0.0, 0.0, 0]}          # [FLOP, ms, launches]
    for k, ev in ops.TIMERS.items():
        p = k.split(":")
        if p[0] != "k" or p[1] not in ("conv1x1_bf16x6", "wgrad_bf16x6", "conv1x1_f32", "wgrad_f32", "sa_mlp_max"):
            continue
        ms_k = sum(a_.elapsed_time(b_) for a_, b_ in ev)
        fl = float(p[5]) * 1e6 * len(ev) if p[1] == "sa_mlp_max" else 2.0 * int(p[2]) * int(p[3]) * int(p[4]) * len(ev)
        acc = pipes["bf16x6" if p[1].endswith("bf16x6") else "f32_mfma"]
        acc[0] += fl; acc[1] += ms_k; acc[2] += len(ev)

    return {"elapsed": elapsed, "per_rank_ms": per_rank_ms, "losses": losses, "pipes": pipes, "allreduce_ms": allreduce_ms, "bucket_bytes": bucket_bytes, "sd": sd, "x_all": x_all, "sp_all": sp_all, "e": e,
            "full": full, "collective_library": collective_library() if world > 1 else None}


def summarize(args, m, world=1):
    """roofline (per pipe) of measure()'s result -> (ms_per_step, roofline dict)."""
    B, T, N = args.batch, args.seq_len, args.num_pts
    full, pipes = m["full"], m["pipes"]
    ms = 1e3 * m["elapsed"] / args.steps
    enc = ENC_GFLOP_PER_SEQ.get((T, N))
    flop = None
    if enc is not None:
        flop = 3.0 * B * enc * 1e9
        if full:
            flop += 3.0 * B * T * N * 4 * args.cnf_steps * CNF_FLOP_PER_POINT_EVAL_DIV
    # priced PER PIPE: the matrix products of the step run on two kernel families with different ceilings -- the bf16x6 kernels
    # (2500 / 6 = 416.7 f32-equivalent TFLOP/s) and the f32-MFMA kernels (157.3) -- so the step's roofline time is
    # FLOP_x6 / 416.7 + FLOP_f32 / 157.3 (measured per launch in the detail pass) and `frac` = that time / the step time
    PEAK_X6 = 2500.0 / 6.0
    fx, ff = pipes["bf16x6"], pipes["f32_mfma"]
    roof_ms = fx[0] / (PEAK_X6 * 1e12) * 1e3 + ff[0] / (PEAK_MFMA_F32_TFLOPS * 1e12) * 1e3

    # HBM-side bytes per step from the committed PMC passes (FETCH_SIZE x 2 + WRITE_SIZE, separate rocprofv3 passes restricted to the
    # matrix / activation kernels by --kernel-include-regex; profiles/kernel_traffic.json), or null
    traffic, traffic_src = None, None
    tpath = os.path.join(ROOT, "profiles", "kernel_traffic.json")
    if os.path.exists(tpath):
        tj = json.load(open(tpath)).get("train_matrix_kernels:%dx%dx%d" % (B, T, N))
        if tj and full:
            traffic = int(1024 * (tj["fetch_size_kb_per_step"] * tj["fetch_correction"] + tj["write_size_kb_per_step"]))
            traffic_src = tj["source"]

    def pipe(v, peak):
        return {"flop_per_step": v[0], "launches_per_step": v[2], "kernel_ms_per_step": round(v[1], 3),
                "achieved": round(v[0] / (v[1] * 1e-3) / 1e12, 3) if v[1] > 0 else None, "peak": round(peak, 1),
                "frac_while_running": round(v[0] / (v[1] * 1e-3) / 1e12 / peak, 4) if v[1] > 0 else None}
    roofline = {"kernel": "every matrix kernel of the step (conv1x1 forward / data gradient, conv1x1_wgrad, fused set abstraction), per pipe",
                "bound": "mfma", "achieved": round((fx[0] + ff[0]) / (ms * 1e-3) / 1e12, 3), "peak": round(PEAK_X6, 1), "unit": "TFLOP/s",
                "frac": round(roof_ms / ms, 4),
                "frac_note": "(FLOP_bf16x6 / 416.7 + FLOP_f32 / 157.3 TFLOP/s) / step time: the share of the step the matrix pipes would need at their peaks",
                "pipes": {"bf16x6": pipe(fx, PEAK_X6), "f32_mfma": pipe(ff, PEAK_MFMA_F32_TFLOPS)},
                "matrix_kernel_ms_per_step": round(fx[1] + ff[1], 3), "traffic": traffic, "traffic_unit": "bytes/step over the matrix / activation kernels of the CNF block and the convs",
                "traffic_source": traffic_src,
                "flop_per_step_model": flop}
    return ms, roofline


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=8, help="sequences per GPU")
    ap.add_argument("--seq-len", type=int, default=10)
    ap.add_argument("--num-pts", type=int, default=1024)
    ap.add_argument("--cnf-steps", type=int, default=8)
    ap.add_argument("--latent-steps", type=int, default=2)
    ap.add_argument("--mode", choices=["full", "pretrain"], default="full")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    # one process per GPU: a plain `python bench_train.py --gpus N` starts its own N ranks (torch.distributed.run on 127.0.0.1);
    # under an external launcher WORLD_SIZE must equal --gpus
    from caspr_amd.utils.launch import ensure_ranks
    rank, local_rank, world = ensure_ranks(args.gpus, __file__, sys.argv[1:], device_count=torch.cuda.device_count)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl")   # RCCL on ROCm
    assert torch.cuda.is_available(), "bench_train.py needs a ROCm GPU (there is no CPU execution path)"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    m = measure(args, dev, rank, world)
    B, T, N = args.batch, args.seq_len, args.num_pts
    full, elapsed, per_rank_ms, losses = m["full"], m["elapsed"], m["per_rank_ms"], m["losses"]
    sd, x_all, sp_all, e = m["sd"], m["x_all"], m["sp_all"], m["e"]
    collective_library = lambda: m["collective_library"]
    if rank == 0:
        ms, roofline = summarize(args, m, world)
        cpu = None
        if not args.no_cpu_baseline:
            from oracle import model as O
            ncores = min(os.cpu_count() or 1, 16)      # profiles/r03_cpu_threads_probe.txt: the intra-op pool peaks at 16 threads on this host
            torch.set_num_threads(ncores)
            s_ = {k: (v.detach().clone().requires_grad_(True) if v.is_floating_point() and not k.endswith(("running_mean", "running_var", "step", "_num_evals")) else v)
                  for k, v in sd.items()}
            xs, sps = x_all[:1], sp_all[:1]
            t1 = time.perf_counter()
            if full:
                l_, _, _ = O.training_loss(s_, xs, sps, e[:T].cpu(), cnf_steps=args.cnf_steps, latent_steps=args.latent_steps)
            else:
                _, tn = O.encode(s_, xs)
                l_ = 100.0 * (tn - sps).abs().mean()
            l_.backward()
            cpu_s = time.perf_counter() - t1
            cpu = {"value": round(1.0 / cpu_s, 5), "unit": "sequences/sec", "cores": ncores, "kind": "port",
                   "sample": "1 sequence (T=%d, N=%d) forward + backward through the oracle's differentiable mode (torch-CPU autograd "
                             "+ C point ops, same RK4 steps), no optimizer step, %.1f s" % (T, N, cpu_s)}
        print(json.dumps({
            "metric": "training sequences/sec (%s step: forward + backward + Adam)" % ("full CaSPR" if full else "T-NOCS pre-training"),
            "value": round(world * B * args.steps / elapsed, 3), "unit": "sequences/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "cfg-3 shard (BASELINE.json configs[2]): run_one_epoch body, B=%d sequences/GPU, T=%d, N=%d, %s; seeded "
                                   "random-init weights" % (B, T, N, "NLL (CNF with Hutchinson divergence) + T-NOCS L1" if full else "T-NOCS L1 only"),
                       "global_batch": world * B, "seq_len": T, "num_pts": N, "cnf_rk4_steps": args.cnf_steps,
                       "latent_rk4_steps": args.latent_steps, "parallelism": "seq-shard x%d + 1 gradient all-reduce" % world,
                       "ranks": {"ms_per_step": per_rank_ms, "collectives": "1 all-reduce of the flat gradient bucket per step (in place: .grad tensors are views of it)",
                                 "allreduce_ms": m["allreduce_ms"], "bucket_bytes": m["bucket_bytes"],
                                 "library": collective_library() if world > 1 else None}},
            "roofline": roofline, "cpu_baseline": cpu, "loss_first": losses[0], "loss_last": losses[-1],
            "max_mem_GB": round(torch.cuda.max_memory_allocated() / 2**30, 1),
        }))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
