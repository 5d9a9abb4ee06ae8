#!/usr/bin/env python
"""Run-to-run determinism of the encoder's pieces at the headline shape (160 frames x 2048 points)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from caspr_amd import ops
from caspr_amd.models import CaSPR
from caspr_amd.utils.synthetic import seeded_state_dict, car_sequences
from oracle import point_ops as P
dev = torch.device("cuda:0")
x, sp = car_sequences(16, 10, 2048, seed=1234)
xyz = x.view(160, 2048, 4)[:, :, :3].contiguous()
xd = xyz.to(dev)
ref = None
for r in range(4):
    idx, new_xyz = ops.furthest_point_sampling(xd, 1024, return_xyz=True)
    torch.cuda.synchronize()
    if ref is None:
        ref = (idx.clone(), new_xyz.clone())
        want = P.furthest_point_sampling(xyz[:8], 1024)
        print("fps vs oracle on 8 frames:", bool(torch.equal(idx[:8].cpu(), want)))
    else:
        print("fps run %d same idx: %s (%d differ)  same xyz: %s" % (r, bool(torch.equal(idx, ref[0])), int((idx != ref[0]).sum()), bool(torch.equal(new_xyz, ref[1]))))
m = CaSPR(check_tol=None)
m.load_state_dict(seeded_state_dict(m.state_dict(), 0))
m = m.to(dev).eval()
xg = x.to(dev)
enc = m.encoder
base = None
for r in range(4):
    with torch.no_grad():
        ind = enc.local_extract.indices(xd)
        z0, tn = m.encode(xg)
    torch.cuda.synchronize()
    cur = {"z0": z0.clone(), "tnocs": tn.clone()}
    for l, d in enumerate(ind["sa"]):
        cur["fps%d" % l] = d["fps_idx"].clone()
        for i, b in enumerate(d["ball_idx"]):
            cur["ball%d_%d" % (l, i)] = b.clone()
    for l, t_ in enumerate(ind["nn"]):
        cur["nn%d" % l] = t_[0].clone()
    if base is None:
        base = cur
    else:
        print("run %d:" % r, {k: ("same" if torch.equal(cur[k], base[k]) else "%d differ" % int((cur[k] != base[k]).sum())) for k in cur}, flush=True)
