#!/usr/bin/env python
"""What the run-time accuracy guard costs at cfg-2, in one process: off / CNF check only / latent check only / both; 10 steps each, 3 rounds.   (GPU)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from caspr_amd import ops
from caspr_amd.models import CaSPR
from caspr_amd.utils.synthetic import seeded_state_dict, car_sequences

dev = torch.device("cuda:0")
m = CaSPR()
m.load_state_dict(seeded_state_dict(m.state_dict(), 0))
m = m.to(dev).eval()
x, sp = car_sequences(16, 10, 2048, seed=1234)
x, ts = x.to(dev), sp[0, :, 0, 3].to(dev)
orig_cnf_begin, orig_lat = m._guard_cnf_begin, m._guard_latent


def run(k=10):
    with torch.no_grad():
        for _ in range(2):
            m.reconstruct(x, num_points=2048, timestamps=ts)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(k):
            m.reconstruct(x, num_points=2048, timestamps=ts)
        torch.cuda.synchronize()
        el = (time.perf_counter() - t0) / k * 1e3
    ops.check_deferred_errors()
    return el


for rnd in range(3):
    for name, tol, cnf, lat in (("off", None, True, True), ("cnf only", 1e-5, True, False), ("latent only", 1e-5, False, True), ("both", 1e-5, True, True)):
        m.check_tol = tol
        m._guard_latent = orig_lat if lat else (lambda *a, **k: None)
        if cnf:
            m._guard_cnf_begin = orig_cnf_begin
        else:
            m._guard_cnf_begin = lambda y, z: None
        print("round %d  guard %-12s: step %.2f ms" % (rnd, name, run()), flush=True)
