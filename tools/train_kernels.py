#!/usr/bin/env python
"""Per-launch HIP-event timing of every matrix kernel of one cfg-3 training step (ops.TIMING = 2), by kernel + shape."""
import os, sys, types
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from caspr_amd import ops
from caspr_amd.models import CaSPR
from caspr_amd.train.loop import train_step
from caspr_amd.utils.synthetic import seeded_state_dict, car_sequences
dev = torch.device("cuda:0")
B, T, N = 8, 10, 1024
m = CaSPR()
m.load_state_dict(seeded_state_dict(m.state_dict(), 0))
m = m.to(dev).train()
opt = torch.optim.Adam(m.parameters(), lr=1e-4)
x, sp = car_sequences(B, T, N)
x, sp = x.to(dev), sp.to(dev)
e = torch.randn(B * T, N, 3, device=dev)
for _ in range(2):
    train_step(m, opt, x, sp, e=e)
torch.cuda.synchronize()
ops.TIMERS.clear()
ops.TIMING = 2
train_step(m, opt, x, sp, e=e)
torch.cuda.synchronize()
ops.TIMING = False
rows = []
for k, ev in ops.TIMERS.items():
    p = k.split(":")
    if p[0] != "k":
        continue
    tot = sum(a.elapsed_time(b) for a, b in ev)
    try:
        fl = float(p[5]) * 1e6 if p[1] == "sa_mlp_max" else 2.0 * int(p[2]) * int(p[3]) * int(p[4])
    except Exception:
        fl = 0.0
    rows.append((tot, k, len(ev), tot / len(ev), fl * len(ev) / tot / 1e9 if tot > 0 else 0))
tot_all = 0
for r in sorted(rows, reverse=True)[:60]:
    tot_all += r[0]
    print("%-46s n %4d  avg %8.3f ms  total %8.3f ms  %7.1f TF" % (r[1], r[2], r[3], r[0], r[4]))
print("listed total %.2f ms; all %.2f ms" % (tot_all, sum(r[0] for r in rows)))
