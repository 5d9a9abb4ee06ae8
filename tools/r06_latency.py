#!/usr/bin/env python
"""Latency of one reconstruct() call on ONE sequence (B = 1, T = 10, N = 2048, 2048 samples), guard on (default) and off: the dependent chain
FPS -> ball query -> set abstraction -> ... -> flow with the chip mostly empty."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from caspr_amd.models import CaSPR
from caspr_amd.utils.synthetic import seeded_state_dict, car_sequences
dev = torch.device("cuda:0")
m = CaSPR(); m.load_state_dict(seeded_state_dict(m.state_dict(), 0)); m = m.to(dev).eval()
x, sp = car_sequences(1, 10, 2048, seed=1234)
x, ts = x.to(dev), sp[0, :, 0, 3].to(dev)
for tol in (1e-5, None):
    m.check_tol = tol
    with torch.no_grad():
        for _ in range(3): m.reconstruct(x, num_points=2048, timestamps=ts)
        torch.cuda.synchronize()
        lat = []
        for _ in range(20):
            t0 = time.perf_counter(); m.reconstruct(x, num_points=2048, timestamps=ts); torch.cuda.synchronize(); lat.append((time.perf_counter() - t0) * 1e3)
    lat.sort()
    print("B = 1 reconstruct(), guard %-5s: median %.2f ms, min %.2f ms (each call synchronised)" % (tol, lat[len(lat) // 2], lat[0]), flush=True)
