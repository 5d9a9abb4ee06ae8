#!/usr/bin/env python
"""Per-launch HIP-event timing of every matrix kernel of one cfg-2 reconstruct() step (ops.TIMING = 2), by kernel + shape:
calls per step, mean ms, f32-equivalent TFLOP/s and the fraction of the pipe's peak.   usage: tools/step_kernels.py [B T N]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from caspr_amd import ops
from caspr_amd.models import CaSPR
from caspr_amd.utils.synthetic import seeded_state_dict, car_sequences
B, T, N = [int(v) for v in sys.argv[1:4]] if len(sys.argv) >= 4 else (16, 10, 2048)
dev = torch.device("cuda:0")
m = CaSPR()
m.load_state_dict(seeded_state_dict(m.state_dict(), 0))
m = m.to(dev).eval()
x, sp = car_sequences(B, T, N)
x, ts = x.to(dev), sp[0, :, 0, 3].to(dev)
with torch.no_grad():
    for _ in range(2):
        m.reconstruct(x, num_points=N, timestamps=ts)
    torch.cuda.synchronize()
    ops.TIMERS.clear()
    ops.TIMING = 2
    K = 3
    for _ in range(K):
        m.reconstruct(x, num_points=N, timestamps=ts)
    torch.cuda.synchronize()
    ops.TIMING = False
rows = []
for k, ev in ops.TIMERS.items():
    p = k.split(":")
    if p[0] != "k":
        continue
    ms = sum(a.elapsed_time(b) for a, b in ev) / len(ev)
    if p[1].startswith("sa_mlp_max"):
        fl, peak = float(p[5]) * 1e6, 157.3
    elif p[1].startswith("conv1x1") or p[1].startswith("wgrad"):
        fl, peak = 2.0 * int(p[2]) * int(p[3]) * int(p[4]), (416.7 if p[1].endswith("bf16x6") else 157.3)
    else:
        continue
    rows.append((ms * len(ev) / K, k, len(ev) / K, ms, fl / ms / 1e9, fl / ms / 1e9 / peak))
tot = 0.0
for r in sorted(rows, reverse=True):
    tot += r[0]
    print("%-44s calls/step %4.1f  %8.3f ms  %7.1f TF  %.3f   (per step %.3f ms)" % (r[1], r[2], r[3], r[4], r[5], r[0]))
print("total %.3f ms per step" % tot)
