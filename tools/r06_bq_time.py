#!/usr/bin/env python
"""Ball query of the first two levels at the headline shape, timed alone: the pair entry (one launch for both scales)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from caspr_amd import lib
if "--lib" in sys.argv:
    lib.SO_PATH = lib.SO_PATH.replace("libcaspr_hip.so", sys.argv[sys.argv.index("--lib") + 1])
    print("library:", os.path.basename(lib.SO_PATH))
from caspr_amd import ops
from caspr_amd.utils.synthetic import car_sequences
x, _ = car_sequences(16, 10, 2048, seed=1234)
xyz = x.view(160, 2048, 4)[:, :, :3].contiguous().cuda()
def t(fn, k=10):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(k): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / k * 1e3
for n, M, ra, rb in ((2048, 1024, 0.02, 0.05), (1024, 512, 0.05, 0.1), (512, 256, 0.1, 0.2)):
    c = xyz[:, :n].contiguous()
    idx, ctr = ops.furthest_point_sampling(c, M, return_xyz=True)
    print("n=%d M=%d: fps %.0f us, ball_query_pair(%g/16, %g/32) %.0f us, single queries %.0f + %.0f us" % (
        n, M, t(lambda: ops.furthest_point_sampling(c, M)), ra, rb, t(lambda: ops.ball_query_pair(ra, 16, rb, 32, c, ctr)),
        t(lambda: ops.ball_query(ra, 16, c, ctr)), t(lambda: ops.ball_query(rb, 32, c, ctr))))
