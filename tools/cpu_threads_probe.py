#!/usr/bin/env python
"""The CPU oracle's reconstruct() on one cfg-2 sequence at several intra-op thread counts: which count should bench.py's
cpu_baseline use on this host?"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from oracle import model as O
from caspr_amd.models import CaSPR
from caspr_amd.utils.synthetic import seeded_state_dict, car_sequences
sd = seeded_state_dict(CaSPR().state_dict(), 0)
x, _ = car_sequences(1, 10, 2048, seed=5)
ts = torch.linspace(0, 1, 10)
y = torch.randn(10, 2048, 3)
print("host cores: %d" % (os.cpu_count() or 0))
for nt in (8, 16, 32, 64, 128, 256):
    if nt > (os.cpu_count() or 1):
        break
    torch.set_num_threads(nt)
    best = 1e9
    for rep in range(2):
        t0 = time.perf_counter()
        O.reconstruct(sd, x, y.view(1, 10, 2048, 3), timestamps=ts, cnf_steps=8, latent_steps=2)
        best = min(best, time.perf_counter() - t0)
    print("threads %3d: %.2f s per sequence -> %.3f sequences/s" % (nt, best, 1.0 / best))
