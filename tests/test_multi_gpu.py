"""RCCL path on real devices: world-size-2 `nccl` run of the only collectives this build uses --
GradBucket.all_reduce_mean (one flat gradient bucket, weighted by shard size) and max_over_ranks (bench timing) -- on
device tensors.  Needs >= 2 visible GPUs; the 1-GPU test box skips it (the same logic runs under gloo in
tests/test_host_cpu.py)."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))

WORKER = r'''
import os, sys, torch, torch.nn as nn, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from caspr_amd.utils.launch import ensure_ranks
from caspr_amd.utils.sharding import max_over_ranks, shard_range
from caspr_amd.train.loop import GradBucket
rank, local_rank, world = ensure_ranks(2, __file__, sys.argv[1:], device_count=torch.cuda.device_count)
dist.init_process_group("nccl")      # RCCL
torch.cuda.set_device(local_rank)
dev = torch.device("cuda", local_rank)
torch.manual_seed(0)
net = nn.Sequential(nn.Linear(8, 16), nn.Tanh(), nn.Linear(16, 1)).to(dev)
x = torch.randn(5, 8, device=dev)                      # 5 "sequences": shards of 3 and 2
full = net(x).pow(2).mean()
want = torch.autograd.grad(full, list(net.parameters()))
lo, hi = shard_range(5, rank, world)
net.zero_grad()
net(x[lo:hi]).pow(2).mean().backward()
bucket = GradBucket(net.parameters())
bucket.all_reduce_mean(weight=hi - lo)
for p, w in zip(net.parameters(), want):
    assert torch.allclose(p.grad, w, atol=1e-6, rtol=1e-5), "weighted bucket all-reduce differs from the global-batch gradient"
assert max_over_ranks(1.0 + rank, dev) == 2.0
dist.barrier()
torch.cuda.synchronize()
if rank == 0:
    print("NCCL_OK")
dist.destroy_process_group()
'''


def test_two_rank_rccl_bucket_and_timing(tmp_path):
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs (RCCL); the gloo twin of this test runs on CPU")
    script = tmp_path / "nccl_worker.py"
    script.write_text(WORKER)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    out = subprocess.run([sys.executable, str(script), ROOT], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0 and "NCCL_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


def _free_parent_cache():
    """The pytest process may hold tens of GB in torch's caching allocator from earlier tests (cfg-3 at size: 64 GB); the ranks started
    below share this device."""
    if torch.cuda.is_available():
        torch.cuda.synchronize()
        torch.cuda.empty_cache()


def _clean_env():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "LOCAL_WORLD_SIZE", "GROUP_RANK")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["MASTER_ADDR"] = "127.0.0.1"
    return env


def _bench_two_ranks(extra):
    import json
    _free_parent_cache()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-sub-blocks",
                          "--no-cpu-baseline", "--no-f32-subblock", "--train-shape", "1,2,1024"] + extra, capture_output=True, text=True, timeout=900, env=_clean_env(), cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["scaling"] == "weak" and r["config"]["global_batch"] == 32
    ranks = r["config"]["ranks"]
    assert len(ranks["ms_per_step"]) == 2 and all(ms > 0 for ms in ranks["ms_per_step"])
    assert abs(r["ms_per_step"] - max(ranks["ms_per_step"])) <= 0.05 * r["ms_per_step"]
    return r, ranks


def _check_sharded_train_leg(r, library):
    """N > 1 (round 6): the line's `train_cfg3` is the SHARDED training step -- the one place the path has a collective (train.py:131-132's
    replacement: one flat gradient all-reduce per step) -- with the collective timed per rank; here at a shrunken per-rank shape."""
    t = r["train_cfg3"]
    assert t is not None and "sharded" in t["workload"] and t["ms_per_step"] > 0 and library in t["ranks"]["library"].lower(), t
    assert len(t["allreduce_ms"]) == 2 and all(v > 0 for v in t["allreduce_ms"]) and t["allreduce_ms_min_over_ranks"] == min(t["allreduce_ms"])
    assert t["bucket_bytes"] >= 4 * 16262189 and len(t["ranks"]["ms_per_step"]) == 2
    assert t["loss_first"] == t["loss_first"] and abs(t["loss_first"]) < 1e6          # finite


def test_bench_two_gpus_is_a_tested_path():
    """`python bench.py --gpus 2` as the driver's scaling run starts it (round-4 review: the first 8-GPU run must not be the first
    execution of this path): one JSON line, n_gpus = 2, the collective library named, one per-rank time per rank, and the whole-job
    value = both ranks' sequences over the slower rank's time."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs (RCCL); test_bench_two_ranks_sharing_one_gpu runs the same rank logic on this box over gloo")
    r, ranks = _bench_two_ranks([])
    _check_sharded_train_leg(r, "ccl")
    assert ranks["library"] and ("nccl" in ranks["library"].lower() or "rccl" in ranks["library"].lower()), ranks
    assert abs(r["value"] - 32 / (r["ms_per_step"] * 1e-3)) <= 0.02 * r["value"]
    per_rank = sum(16 / (ms * 1e-3) for ms in ranks["ms_per_step"])
    assert 0.8 * per_rank <= r["value"] <= 1.01 * per_rank, (r["value"], per_rank)
    assert r["roofline"]["frac"] > 0.3


def test_bench_two_ranks_sharing_one_gpu():
    """The same command with --share-gpu: two ranks on ONE device over gloo.  Measures nothing (the line says so and carries no
    value) but executes everything `bench.py --gpus N` does per rank -- the self-launch under torch.distributed.run, contiguous
    sequence blocks, barriers, max-over-ranks and per-rank times, rank 0's single line, the return-code broadcast -- on the box the
    GPU suite actually runs on."""
    r, ranks = _bench_two_ranks(["--share-gpu"])
    assert r["value"] is None and "NOT a measurement" in r["test_mode"] and "gloo" in ranks["library"]
    _check_sharded_train_leg(r, "gloo")


MODEL_WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from caspr_amd.utils.launch import ensure_ranks
from caspr_amd.utils.sharding import shard_range
from caspr_amd.utils.synthetic import seeded_state_dict, car_sequences
from caspr_amd.train.loop import GradBucket, training_loss, broadcast_model
from caspr_amd.models import CaSPR
backend = sys.argv[2]                # "nccl" (RCCL, one GPU per rank) | "gloo" (the ranks share the visible devices)
rank, local_rank, world = ensure_ranks(2, __file__, sys.argv[1:], device_count=torch.cuda.device_count if backend == "nccl" else None)
dist.init_process_group(backend)
local_rank = local_rank % torch.cuda.device_count()
torch.cuda.set_device(local_rank)
dev = torch.device("cuda", local_rank)
B, T, N = 4, 2, 1024
x, sp = car_sequences(B, T, N, seed=11)
torch.manual_seed(3)
e = torch.randn(B * T, N, 3)

def grads(xs, sps, es, bucket_weight=None):
    m = CaSPR(cnf_rk4_steps=4, latent_rk4_steps=2)
    m.load_state_dict(seeded_state_dict(m.state_dict(), 0))
    m = m.to(dev).train()
    bucket = GradBucket(m.parameters()) if bucket_weight is not None else None
    if bucket is not None:
        bucket.zero()
    loss, _, _ = training_loss(m(xs.to(dev), sps.to(dev), e=es.to(dev)), 0.01, 100.0)
    loss.backward()
    if bucket is not None:
        bucket.all_reduce_mean(weight=bucket_weight)
        bucket.drop_untouched()
    return {n: (p.grad.detach().clone() if p.grad is not None else None) for n, p in m.named_parameters()}, float(loss)

full, loss_full = grads(x, sp, e)                                   # every rank: the 1-rank gradient of the 4-sequence batch
lo, hi = shard_range(B, rank, world)
mine, loss_mine = grads(x[lo:hi], sp[lo:hi], e[lo * T:hi * T], bucket_weight=hi - lo)     # the sharded step: backward + ONE RCCL all-reduce
num = den = 0.0
per = []
# the bias of a conv in front of GroupNorm(16, 16) (one channel per group: the first set-abstraction scale's first two layers,
# pointnet2.py:649-703) cannot change the output: its gradient is mathematically zero and what f32 arithmetic returns for it is rounding
# noise of whatever order the batch was summed in (tests/test_hip_train_cfg3.py records it the same way)
zero_grad = {"encoder.local_extract.set_abstractions.0.pointnet_modules.0.conv_layers.%d.bias" % i for i in (0, 1)}
for n, g in full.items():
    h = mine[n]
    assert (g is None) == (h is None), "parameter %s: touched in one run only" % n
    if g is not None and n not in zero_grad:
        d2 = float((g.double() - h.double()).pow(2).sum())
        num += d2
        den += float(g.double().pow(2).sum())
        per.append((d2, n, float(g.double().pow(2).sum())))
rel = (num / den) ** 0.5
worst = ", ".join("%s %.2e of its norm" % (n, (d2 / max(g2, 1e-300)) ** 0.5) for d2, n, g2 in sorted(per, reverse=True)[:6])
assert rel <= 2e-4, "rank %d: |sharded - global| / |global| = %.3e over the whole gradient (loss %r vs shard %r); largest contributions: %s" % (rank, rel, loss_full, loss_mine, worst)
dist.barrier()
torch.cuda.synchronize()
if rank == 0:
    print("MODEL_RANKS_OK rel=%.3e" % rel)
dist.destroy_process_group()
'''


def _real_model_two_ranks(tmp_path, backend):
    _free_parent_cache()
    script = tmp_path / "model_worker.py"
    script.write_text(MODEL_WORKER)
    out = subprocess.run([sys.executable, str(script), ROOT, backend], capture_output=True, text=True, timeout=900, env=_clean_env())
    assert out.returncode == 0 and "MODEL_RANKS_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]


def test_two_rank_train_step_of_the_real_model(tmp_path):
    """The sharded training step of the REAL CaSPR on RCCL: two ranks with two sequences each (T = 2, N = 1024), backward on the HIP
    kernels, ONE all-reduce of the flat gradient bucket -- against the one-rank gradient of the four-sequence batch, to the
    shard-identity bound of tests/test_hip_train_cfg3.py (2e-4 of the gradient's norm: only the f32 reduction order differs)."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs (RCCL); test_two_rank_train_step_of_the_real_model_sharing_one_gpu runs it on this box over gloo")
    _real_model_two_ranks(tmp_path, "nccl")


def test_two_rank_train_step_of_the_real_model_sharing_one_gpu(tmp_path):
    """The same two-rank step with both ranks on ONE device and gloo carrying the all-reduce: every line of the sharded path except
    the transport runs on the box the GPU suite actually runs on (tests/test_host_cpu.py holds the CPU twin on the oracle model)."""
    _real_model_two_ranks(tmp_path, "gloo")
