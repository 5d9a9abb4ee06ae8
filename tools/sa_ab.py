#!/usr/bin/env python
"""The fused set-abstraction calls of one cfg-2 encoder pass, per (level, scale), on this tree's library and on a library whose
sa_mlp.hip is the previous commit's (built by hand next to it: libcaspr_hip_prevsa.so = HEAD~'s sa_mlp.hip + this tree's other
objects), one scale at a time, alternating, in one box.   (GPU)"""
import os, sys, subprocess
if "--child" not in sys.argv:
    for libname, lo in (("libcaspr_hip.so", "1"), ("libcaspr_hip_prevsa.so", "1"), ("libcaspr_hip.so", "1"), ("libcaspr_hip_prevsa.so", "1")):
        subprocess.run([sys.executable, os.path.abspath(__file__), "--child", libname, lo])
    sys.exit(0)
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from caspr_amd import lib
lib.SO_PATH = lib.SO_PATH.replace("libcaspr_hip.so", sys.argv[2])
import torch
from caspr_amd import ops
from caspr_amd.models import CaSPR
import caspr_amd.models.pointnet2 as P2
from caspr_amd.utils.synthetic import seeded_state_dict, car_sequences
P2.SCALE_STREAMS = False
P2.LO_PARTS = sys.argv[3] == "1"
dev = torch.device("cuda:0")
m = CaSPR()
m.load_state_dict(seeded_state_dict(m.state_dict(), 0))
m = m.to(dev).eval()
x, _ = car_sequences(16, 10, 2048, seed=1234)
x = x.to(dev)
with torch.no_grad():
    for _ in range(2):
        m.encode(x)
    torch.cuda.synchronize()
    ops.TIMERS.clear()
    ops.TIMING = 2
    for _ in range(5):
        m.encode(x)
    torch.cuda.synchronize()
    ops.TIMING = False
row = []
tot = 0.0
for k, v in ops.TIMERS.items():
    if k.startswith("k:sa_mlp_max"):
        ms = sum(a.elapsed_time(b) for a, b in v) / 5
        tot += ms
        row.append("%s %.3f" % (":".join(k.split(":")[2:4]), ms))
print("%-24s total %.3f ms | %s" % (sys.argv[2], tot, "  ".join(row)), flush=True)
