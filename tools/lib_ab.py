#!/usr/bin/env python
"""cfg-2 reconstruct() on this tree's library and on another build of it next to it (csrc/<name>.so: one kernel changed by hand), separate
processes alternating on one box, 20 steps each.   usage: tools/lib_ab.py libcaspr_hip_prev.so   (GPU)"""
import os, sys, subprocess, time
if "--child" not in sys.argv:
    other = sys.argv[1]
    for rnd in range(3):
        for name in ("libcaspr_hip.so", other):
            subprocess.run([sys.executable, os.path.abspath(__file__), "--child", name])
    sys.exit(0)
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from caspr_amd import lib
lib.SO_PATH = lib.SO_PATH.replace("libcaspr_hip.so", sys.argv[2])
import torch
from caspr_amd.models import CaSPR
from caspr_amd.utils.synthetic import seeded_state_dict, car_sequences
dev = torch.device("cuda:0")
m = CaSPR()
m.load_state_dict(seeded_state_dict(m.state_dict(), 0))
m = m.to(dev).eval()
x, sp = car_sequences(16, 10, 2048, seed=1234)
x, ts = x.to(dev), sp[0, :, 0, 3].to(dev)
with torch.no_grad():
    for _ in range(3):
        m.reconstruct(x, num_points=2048, timestamps=ts)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        m.reconstruct(x, num_points=2048, timestamps=ts)
    torch.cuda.synchronize()
print("%-28s step %.2f ms" % (sys.argv[2], (time.perf_counter() - t0) / 20 * 1e3), flush=True)
