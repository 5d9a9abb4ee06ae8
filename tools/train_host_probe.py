#!/usr/bin/env python
"""Is the training step bound by the host?  Time until train_step() returns (everything enqueued) against the time until the GPU is
done, at cfg-3's per-GPU shard (8 x 10 x 1024)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from caspr_amd.models import CaSPR
from caspr_amd.train.loop import train_step
from caspr_amd.utils.synthetic import car_sequences, seeded_state_dict
dev = torch.device("cuda:0")
m = CaSPR(); m.load_state_dict(seeded_state_dict(m.state_dict(), 0)); m = m.to(dev).train()
opt = torch.optim.Adam(m.parameters(), lr=1e-4)
x, sp = car_sequences(8, 10, 1024, seed=1234)
x, sp = x.to(dev), sp.to(dev)
e = torch.randn(80, 1024, 3, device=dev)
for _ in range(2):
    train_step(m, opt, x, sp, e=e)
torch.cuda.synchronize()
from caspr_amd.train.loop import training_loss
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    opt.zero_grad()
    losses = m(x, sp, e=e)
    loss, _, _ = training_loss(losses, 0.01, 100.0)
    t1 = time.perf_counter()
    loss.backward()
    t2 = time.perf_counter()
    opt.step()
    t3 = time.perf_counter()
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    print("host: forward enqueued at %.1f ms, backward at %.1f, optimizer at %.1f; GPU done at %.1f ms" % tuple((t - t0) * 1e3 for t in (t1, t2, t3, t4)))
