#!/usr/bin/env python
"""The point-CNF sampling solve alone at the cfg-2 size (160 frames x 2048 points, 8 RK4 steps), a few launches: the workload
for rocprofv3 passes on the dominant kernel (tools/profile_cnf.sh)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from caspr_amd.models import CaSPR
from caspr_amd.utils.synthetic import seeded_state_dict
dev = torch.device("cuda:0")
m = CaSPR(); m.load_state_dict(seeded_state_dict(m.state_dict(), 0)); m = m.to(dev).eval()
BT, n = 160, 2048
g = torch.Generator().manual_seed(5)
y, c = torch.randn(BT, n, 3, generator=g).to(dev), torch.randn(BT, 1600, generator=g).to(dev)
k = int(sys.argv[1]) if len(sys.argv) > 1 else 3
with torch.no_grad():
    m.point_cnf(y, c, reverse=True); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(k): m.point_cnf(y, c, reverse=True)
    b.record(); torch.cuda.synchronize()
print("point_cnf (hyper conv + solve): %.2f ms" % (a.elapsed_time(b) / k))
