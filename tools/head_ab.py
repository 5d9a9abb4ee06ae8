#!/usr/bin/env python
"""A/B in one process on one box: the last head layer's 64-channel remainder beside the main tiles (models/tpointnet2.py: TAIL_BESIDE) and the
T-NOCS regression joined behind the flow's launch (models/caspr.py: JOIN_TNOCS_LATE) and queued right in front of it (models/tpointnet2.py: LATE_TNOCS_LAUNCH); cfg-2 reconstruct(), 10 steps each, 3 rounds; outputs
compared bit for bit with the serial order.   (GPU)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from caspr_amd.models import CaSPR
import caspr_amd.models.tpointnet2 as TP
import caspr_amd.models.caspr as C
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
        t0 = time.perf_counter()
        for _ in range(k):
            m.reconstruct(x, num_points=2048, timestamps=ts)
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k * 1e3, o


ref = None
for rnd in range(3):
    for tail, late, launch in ((False, False, False), (True, False, False), (True, True, False), (True, True, True)):
        TP.TAIL_BESIDE, C.JOIN_TNOCS_LATE, TP.LATE_TNOCS_LAUNCH = tail, late, launch
        el, o = run()
        if ref is None:
            ref = o
        same = torch.equal(o[2], ref[2]) and torch.equal(o[3], ref[3])
        print("round %d  remainder beside the tiles %-5s  T-NOCS joined late %-5s  queued in front of the flow's launch %-5s : step %.2f ms   outputs %s"
              % (rnd, tail, late, launch, el, "identical" if same else "DIFFER"), flush=True)
