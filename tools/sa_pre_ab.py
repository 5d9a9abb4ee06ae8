#!/usr/bin/env python
"""A/B in one process on one box: the wide set-abstraction levels with the first layer pre-aggregated (models/pointnet2.py: PRE_AGGREGATE --
its feature part once per source point) against the fused gather + three layers; cfg-2 reconstruct(), 10 steps per figure, alternating.  (GPU)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from caspr_amd import ops
from caspr_amd.models import CaSPR
import caspr_amd.models.pointnet2 as P2
from caspr_amd.utils.synthetic import seeded_state_dict, car_sequences

dev = torch.device("cuda:0")
m = CaSPR()
m.load_state_dict(seeded_state_dict(m.state_dict(), 0))
m = m.to(dev).eval()
x, sp = car_sequences(16, 10, 2048, seed=1234)
x, ts = x.to(dev), sp[0, :, 0, 3].to(dev)
torch.manual_seed(2)
yb = torch.randn(16, 10, 2048, 3).to(dev)


def run(k=10):
    with torch.no_grad():
        for _ in range(2):
            o = m.reconstruct(x, num_points=2048, timestamps=ts, y=yb)
        torch.cuda.synchronize()
        ops.TIMERS.clear()
        ops.TIMING = True
        ops.TIMING_ONLY = None
        t0 = time.perf_counter()
        for _ in range(k):
            m.reconstruct(x, num_points=2048, timestamps=ts)
        torch.cuda.synchronize()
        el = (time.perf_counter() - t0) / k * 1e3
        ops.TIMING = False
    st = {n: sum(a.elapsed_time(b) for a, b in v) / k for n, v in ops.TIMERS.items() if n in ("enc_set_abstraction", "enc_local_pointnet2")}
    return el, st, o


ref = None
for rnd in range(3):
    for on in (True, False):
        P2.PRE_AGGREGATE = on
        el, st, o = run()
        if ref is None:
            ref = o
        print("round %d  first layer pre-aggregated %-5s : step %.2f ms   set abstraction (wall) %.2f ms   max |x - x_first| %.2e  max |tnocs - tnocs_first| %.2e"
              % (rnd, on, el, st.get("enc_set_abstraction", 0.0), float((o[2] - ref[2]).abs().max()), float((o[3] - ref[3]).abs().max())), flush=True)
