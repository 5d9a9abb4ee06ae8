#!/usr/bin/env python
"""A/B in ONE process on ONE box (the boxes of the pool differ by 3-5 %): the cfg-2 reconstruct() step with the two scales of a
set-abstraction level on two streams or one (models/pointnet2.py: SCALE_STREAMS), with the f64 re-evaluation of each scale's small balls
on a stream of its own beside the MFMA kernel or behind it (F64_STREAMS; round 5, second part), and with / without the low parts between
the first two levels (LO_PARTS); alternating, three rounds of 10 steps each; the outputs of every variant compared bit for bit.   (GPU)"""
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


def run(k=10):
    with torch.no_grad():
        for _ in range(2):
            m.reconstruct(x, num_points=2048, timestamps=ts)
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
    return el, st


torch.manual_seed(2)
yb = torch.randn(16, 10, 2048, 3).to(dev)
ref = None
for rnd in range(3):
    for streams, f64s, lo in ((True, True, True), (True, False, True), (False, False, True), (True, True, True), (True, False, True)):
        P2.SCALE_STREAMS, P2.F64_STREAMS, P2.LO_PARTS = streams, f64s, lo
        with torch.no_grad():
            o = m.reconstruct(x, num_points=2048, timestamps=ts, y=yb)
        torch.cuda.synchronize()
        if ref is None:
            ref = o
        same = torch.equal(o[2], ref[2]) and torch.equal(o[3], ref[3])
        el, st = run()
        print("round %d  scale streams %-5s  f64 halves on their own streams %-5s  lo parts %-5s : step %.2f ms   set abstraction (wall) %.2f ms   local branch %.2f ms   outputs %s"
              % (rnd, streams, f64s, lo, el, st.get("enc_set_abstraction", 0.0), st.get("enc_local_pointnet2", 0.0), "identical" if same else "DIFFER"), flush=True)
