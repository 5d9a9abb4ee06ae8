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
