#!/usr/bin/env python
"""The CPU-generator draw of the base samples (983,040 normals per cfg-2 step, models/utils.py:25) with 1 and 8 processes at once: what 8
ranks per node cost each other on the host (round-3 review, weak #10).  The draw sits under the encoder (~24 ms of GPU time)."""
import torch, time, sys, multiprocessing as mp
def work(q, n):
    torch.set_num_threads(1)
    buf = torch.empty(160, 2048, 3)
    torch.randn(160, 2048, 3, out=buf)
    t = time.perf_counter()
    for _ in range(n): torch.randn(160, 2048, 3, out=buf)
    q.put((time.perf_counter() - t) / n * 1e3)
if __name__ == "__main__":
    for procs in (1, 8):
        q = mp.Queue(); ps = [mp.Process(target=work, args=(q, 50)) for _ in range(procs)]
        [p.start() for p in ps]; r = [q.get() for _ in ps]; [p.join() for p in ps]
        print("%d concurrent draws of 983,040 normals: %.2f ms each (min %.2f, max %.2f)" % (procs, sum(r) / len(r), min(r), max(r)))
