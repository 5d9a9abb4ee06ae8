#!/usr/bin/env python
"""How far must the f64 re-evaluation of small neighbourhoods (csrc/sa_mlp.hip: sa_repair_f64_kernel) reach?  Debug flavour of the
library, CASPR_SA_REPAIR_K = -1 (off) / 4 / 8: cfg-5's T-NOCS / xyz error against the f64 oracle on sequence 0 of (1, 20, 4096) i.i.d.
clouds, and the set-abstraction kernel time of one cfg-2 encoder pass (car clouds, 16 x 10 x 2048) and one cfg-5 pass.   (GPU)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from caspr_amd import lib
lib.SO_PATH = lib.SO_PATH.replace("libcaspr_hip.so", "libcaspr_hip_debug.so")
import torch
from caspr_amd import ops
from caspr_amd.models import CaSPR
from caspr_amd.utils.synthetic import seeded_state_dict, random_clouds, car_sequences
from oracle import model as O

dev = torch.device("cuda:0")
m = CaSPR(cnf_rk4_steps=8, latent_rk4_steps=4)
sd = seeded_state_dict(m.state_dict(), 0)
m.load_state_dict(sd)
m = m.to(dev).eval()
T, N = 20, 4096
sd64 = {k: v.double() for k, v in sd.items()}
torch.set_num_threads(32)
big = random_clouds(64, T, N, seed=1234)
cases = {"seed 7": random_clouds(2, T, N, seed=7)[:1], "seed 77": random_clouds(1, T, N, seed=77), "bench seq 0": big[:1], "bench seq 63": big[63:]}
ts = cases["seed 7"][0, :, 0, 3] / 5.0
torch.manual_seed(5)
yb = torch.randn(1, T, 256, 3)
ref = {}
t0 = time.time()
for name, xx in cases.items():
    _, _, x64, t64 = O.reconstruct(sd64, xx.double(), yb.double(), timestamps=ts.double(), cnf_steps=8, latent_steps=4)
    ref[name] = t64
print("f64 oracle: %.0f s" % (time.time() - t0), flush=True)
xc, _ = car_sequences(16, 10, 2048, seed=1234)
xc = xc.to(dev)
x5b = random_clouds(8, T, N, seed=1234).to(dev)


def sa_ms(x):
    with torch.no_grad():
        m.encode(x)
        torch.cuda.synchronize()
        ops.TIMERS.clear()
        ops.TIMING = 2
        for _ in range(3):
            m.encode(x)
        torch.cuda.synchronize()
        ops.TIMING = False
    tot = 0.0
    for k, v in ops.TIMERS.items():
        if k.startswith("k:sa_mlp_max"):
            tot += sum(a.elapsed_time(b) for a, b in v)
    return tot / 3


for rk, rk1 in (("-1", "0"), ("4", "0"), ("0", "0"), ("0", "ns16")):
    os.environ["CASPR_SA_REPAIR_K"] = rk
    os.environ["CASPR_SA_REPAIR_K1"] = "0"
    os.environ["CASPR_SA_WIDE_NS16_ONLY"] = "1" if rk1 == "ns16" else "0"
    errs = []
    for name, xx in cases.items():
        with torch.no_grad():
            _, _, gx, gt = m.reconstruct(xx.to(dev), num_points=256, timestamps=ts.to(dev), y=yb.to(dev))
        errs.append("%s %.2e" % (name, float((gt.cpu().double() - ref[name]).abs().max())))
    print("CASPR_SA_REPAIR_K=%2s (-1 off, 4: K <= 4, 0: K <= 8 at level 0 = production) K1=%s (8: K <= 8 at level 1 too): cfg5 (1,20,4096) T-NOCS vs f64: %s | set abstraction: cfg-2 cars %.3f ms, cfg-5 (8 seq) %.3f ms"
          % (rk, rk1, ", ".join(errs), sa_ms(xc), sa_ms(x5b)), flush=True)
