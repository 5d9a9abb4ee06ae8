"""One process per GPU, started from a plain `python script.py --gpus N`.

The reference's only parallelism is nn.DataParallel inside one process (caspr/train.py:131-132).  This build runs one
process per GPU over RCCL; the driver may start the ranks itself (`python -m torch.distributed.run
--nproc-per-node N script.py --gpus N`) or call the script directly -- in which case `ensure_ranks` re-executes the same
command line under torch.distributed.run on 127.0.0.1 and returns only inside the ranks.  A line whose `--gpus` disagrees
with the number of ranks that actually ran is never printed: `ensure_ranks` raises instead.
"""
import os
import socket
import subprocess
import sys


def _free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def ensure_ranks(n_ranks, script, argv, device_count=None):
    """Return (rank, local_rank, world) with world == n_ranks, launching the ranks if nobody did.

    * WORLD_SIZE set (we are a rank of somebody's launcher): it must equal n_ranks.
    * WORLD_SIZE unset and n_ranks == 1: single process, returns (0, 0, 1).
    * WORLD_SIZE unset and n_ranks > 1: runs `python -m torch.distributed.run --nnodes=1 --nproc-per-node n_ranks
      --master-addr 127.0.0.1 --master-port <free> script argv...`, waits, and exits with its return code.
    `device_count` (callable) guards against asking for more ranks than visible GPUs."""
    if n_ranks < 1:
        raise ValueError("--gpus must be >= 1, got %d" % n_ranks)
    if "WORLD_SIZE" in os.environ:
        world = int(os.environ["WORLD_SIZE"])
        if world != n_ranks:
            raise RuntimeError("--gpus %d but the launcher started %d rank(s) (WORLD_SIZE): refusing to report a line whose "
                               "n_gpus is not the number of GPUs that ran" % (n_ranks, world))
        return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), world
    if n_ranks == 1:
        return 0, 0, 1
    if device_count is not None:
        have = int(device_count())
        if have < n_ranks:
            raise RuntimeError("--gpus %d but only %d device(s) are visible" % (n_ranks, have))
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL needs it on this driver
    env["MASTER_ADDR"] = "127.0.0.1"
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 1) // n_ranks)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % n_ranks,
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(script)] + list(argv)
    sys.stdout.flush()
    sys.stderr.flush()
    sys.exit(subprocess.call(cmd, env=env))
