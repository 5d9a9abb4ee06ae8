#!/usr/bin/env python
"""Are the headline call's outputs the same bits whatever the encoder's schedule (pointnet2.BALL_QUERY_PAIR, LONG_SCALE_ON_MAIN), and run to run?"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from caspr_amd.models import CaSPR
from caspr_amd.models import pointnet2 as P2
from caspr_amd.utils.synthetic import seeded_state_dict, car_sequences
dev = torch.device("cuda:0")
m = CaSPR(check_tol=None)
m.load_state_dict(seeded_state_dict(m.state_dict(), 0))
m = m.to(dev).eval()
x, sp = car_sequences(16, 10, 2048, seed=1234)
x, ts = x.to(dev), sp[0, :, 0, 3].to(dev)
torch.manual_seed(0); y = torch.randn(16, 10, 2048, 3, device=dev)
def outs():
    with torch.no_grad():
        z0, tn = m.encode(x)
        o = m.reconstruct(x, num_points=2048, timestamps=ts, y=y)
    torch.cuda.synchronize()
    return {"z0": z0.clone(), "tnocs_enc": tn.clone(), "x": o[2].clone(), "tnocs": o[3].clone()}
base = None
for pair, long_main in ((False, False), (False, False), (True, False), (True, False), (False, True), (False, True), (True, True), (True, True)):
    P2.BALL_QUERY_PAIR, P2.LONG_SCALE_ON_MAIN = pair, long_main
    o = outs()
    if base is None:
        base = o
    print("pair %-5s long_on_main %-5s :" % (pair, long_main), {k: ("same" if torch.equal(o[k], base[k]) else "max |diff| %.3e in %d entries" % (float((o[k] - base[k]).abs().max()), int((o[k] != base[k]).sum()))) for k in o}, flush=True)
