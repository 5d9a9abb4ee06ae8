#!/usr/bin/env python
"""Full-chip consistency check of the bf16x6 sampling kernel against the f32-MFMA kernel (production library): same solve at
BT frames x n points, where do the two differ?"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from caspr_amd import ops
from caspr_amd.models import CaSPR
from caspr_amd.utils.synthetic import seeded_state_dict
dev = torch.device("cuda:0")
m = CaSPR(); m.load_state_dict(seeded_state_dict(m.state_dict(), 0)); m = m.to(dev).eval()
BT, n = int(sys.argv[1]) if len(sys.argv) > 1 else 160, int(sys.argv[2]) if len(sys.argv) > 2 else 2048
scale = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
g = torch.Generator().manual_seed(5)
y, c = torch.randn(BT, n, 3, generator=g).to(dev), (scale * torch.randn(BT, 1600, generator=g)).to(dev)
with torch.no_grad():
    ops.set_matmul_mode(cnf=False); ref = m.point_cnf(y, c, reverse=True)
    ops.set_matmul_mode(cnf=True)
    for rep in range(3):
        got = m.point_cnf(y, c, reverse=True)
        d = (got - ref).abs().amax(dim=2)                       # (BT, n)
        bad = d > 1e-4
        print("rep %d: max |x6 - f32| %.3e   points > 1e-4: %d of %d   |ref|max %.2f   nan %d" % (rep, float(d.max()), int(bad.sum()), d.numel(), float(ref.abs().max()), int(torch.isnan(got).sum())))
        if bad.any():
            fr = bad.any(dim=1).nonzero().flatten().tolist()
            print("  frames with bad points: %d (first %s)" % (len(fr), fr[:10]))
            pts = bad.nonzero()[:20].tolist()
            print("  first bad (frame, point): %s" % pts)
            wg = torch.div(bad.nonzero()[:, 1], 128, rounding_mode="floor")
            print("  bad points by 32-point wave slot within the workgroup: %s" % torch.bincount((bad.nonzero()[:, 1] % 128) // 32, minlength=4).tolist())
            print("  largest per-frame max errors: %s" % [round(v, 4) for v in d.amax(dim=1).topk(min(8, BT)).values.tolist()])
