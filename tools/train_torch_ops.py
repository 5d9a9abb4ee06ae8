#!/usr/bin/env python
"""Which torch (ATen) operators the cfg-3 training step launches around the HIP kernels: count and GPU time per operator and input shape."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from torch.profiler import profile, ProfilerActivity
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
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    train_step(m, opt, x, sp, e=e)
    torch.cuda.synchronize()
rows = []
for ev in prof.key_averages(group_by_input_shape=True):
    t = getattr(ev, "self_device_time_total", None)
    if t is None:
        t = getattr(ev, "self_cuda_time_total", 0)
    if t > 0:
        rows.append((t / 1e3, ev.count, ev.key, str(ev.input_shapes)[:90]))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows if r[2].startswith("aten::"))
print("aten operators with GPU time: %.2f ms in %d calls" % (tot, sum(r[1] for r in rows if r[2].startswith("aten::"))))
for r in rows[:70]:
    print("%8.3f ms  n=%5d  %-34s %s" % r)
