"""Time one training step (forward + backward + Adam) on synthetic sequences: T-NOCS pre-training (default) or the full
model (`full` as 5th argument: loss = 0.01*mean(sum_n nll) + 100*mean(L1 tnocs), train_utils.py:151-165).
usage: PYTHONPATH=. python tools/train_bench.py [B T N steps [full]]      (cfg-3 per GPU: 8 10 1024)"""
import json
import sys
import time

import torch

from caspr_amd.models import CaSPR
from caspr_amd.utils.synthetic import car_sequences, seeded_state_dict

B, T, N, steps = (int(a) for a in (sys.argv[1:5] + ["8", "10", "1024", "5"][len(sys.argv[1:5]):]))
full = len(sys.argv) > 5 and sys.argv[5] == "full"
dev = torch.device("cuda:0")
m = CaSPR(pretrain_tnocs=not full)
sd = seeded_state_dict(CaSPR().state_dict(), 0)
m.load_state_dict(sd if full else {k: v for k, v in sd.items() if k.startswith("encoder.")})
m = m.to(dev).train()
opt = torch.optim.Adam(m.parameters(), lr=1e-4)
x, sp = car_sequences(B, T, N, seed=1234)
x, sp = x.to(dev), sp.to(dev)


def step():
    opt.zero_grad()
    out = m(x, sp)
    loss = 100.0 * out[-1][:, :, :, :4].mean()
    if full:
        loss = loss + 0.01 * out[0].sum(2).mean()
    loss.backward()
    opt.step()
    return loss


for _ in range(2):
    l0 = step()
torch.cuda.synchronize()
t0 = time.time()
for _ in range(steps):
    l = step()
torch.cuda.synchronize()
dt = (time.time() - t0) / steps
# forward only (taped) for the split
torch.cuda.synchronize()
t1 = time.time()
for _ in range(steps):
    with torch.enable_grad():
        out = m(x, sp)
torch.cuda.synchronize()
df = (time.time() - t1) / steps
print(json.dumps({"mode": "full" if full else "pretrain_tnocs", "B": B, "T": T, "N": N, "ms_per_step": dt * 1e3, "ms_taped_forward": df * 1e3, "sequences_per_s": B / dt,
                  "loss_first": float(l0), "loss_last": float(l), "max_mem_GB": torch.cuda.max_memory_allocated() / 2**30}))
