"""Sequence sharding for multi-GPU runs: one process per GPU, contiguous blocks of
sequences per rank, NO data-path collective for inference (every stage is independent across the
batch index once the integrators are fixed-step).  Collectives are used only for timing / reporting."""
import torch
import torch.distributed as dist


def shard_range(total, rank, world):
    """Contiguous [lo, hi) block of `total` sequences owned by `rank` (sizes differ by at most one)."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _coll_device(device):
    """Where the small bookkeeping collectives of this module run: on `device` under RCCL ("nccl"), on the host under gloo (whose
    all_gather does not take device tensors; bench.py --share-gpu and the CPU tests)."""
    if dist.is_available() and dist.is_initialized() and dist.get_backend() == "gloo":
        return torch.device("cpu")
    return device


def max_over_ranks(value, device):
    """Max of a python float over all ranks (used for the bench's max-over-ranks step time)."""
    device = _coll_device(device)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, device):
    device = _coll_device(device)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def gather_sharded(local, total, rank, world):
    """All-gather per-rank result rows (dim 0) back into global order (tests / metric collection only)."""
    if not (dist.is_available() and dist.is_initialized()) or world == 1:
        return local
    sizes = [shard_range(total, r, world)[1] - shard_range(total, r, world)[0] for r in range(world)]
    mx = max(sizes)
    pad = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    outs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(outs, pad)
    return torch.cat([o[:s] for o, s in zip(outs, sizes)], dim=0)


def per_rank_values(value, device):
    """Every rank's python float, in rank order (self-diagnosing multi-GPU bench lines: which rank was the slow one)."""
    device = _coll_device(device)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    if not (dist.is_available() and dist.is_initialized()):
        return [float(value)]
    outs = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(outs, t)
    return [float(o.item()) for o in outs]


def collective_library():
    """What torch.distributed's "nccl" backend is on this build: RCCL's version on ROCm."""
    try:
        v = torch.cuda.nccl.version()
        return "RCCL %s (torch.distributed backend \"nccl\", HIP %s)" % (".".join(str(i) for i in v) if isinstance(v, tuple) else v, torch.version.hip)
    except Exception as ex:      # pragma: no cover
        return "unknown (%s)" % ex
