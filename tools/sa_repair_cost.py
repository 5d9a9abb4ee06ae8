#!/usr/bin/env python
"""Cost of the f64 re-evaluation of small neighbourhoods per (level, scale) at cfg-2 (car clouds, 16 x 10 x 2048): the fused
set-abstraction call timed with CASPR_SA_REPAIR_K = -1 (off) / 4 (K <= 4) / 0 (K <= 8), debug library, and how many neighbourhoods fall
into each range.   (GPU)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from caspr_amd import lib
lib.SO_PATH = lib.SO_PATH.replace("libcaspr_hip.so", sys.argv[sys.argv.index("--lib") + 1] if "--lib" in sys.argv else "libcaspr_hip_debug.so")
import torch
from caspr_amd import ops
from caspr_amd.models import CaSPR
from caspr_amd.utils.synthetic import seeded_state_dict, car_sequences

import caspr_amd.models.pointnet2 as P2
P2.SCALE_STREAMS = False          # one scale at a time: a launch's own duration, not its share of an overlap
dev = torch.device("cuda:0")
m = CaSPR()
m.load_state_dict(seeded_state_dict(m.state_dict(), 0))
m = m.to(dev).eval()
x, _ = car_sequences(16, 10, 2048, seed=1234)
x = x.to(dev)
res = {}
with torch.no_grad():
    for rk in ("-1", "4", "0"):
        os.environ["CASPR_SA_REPAIR_K"] = rk
        m.encode(x)
        torch.cuda.synchronize()
        ops.TIMERS.clear()
        ops.TIMING = 2
        for _ in range(3):
            m.encode(x)
        torch.cuda.synchronize()
        ops.TIMING = False
        for k, v in ops.TIMERS.items():
            if k.startswith("k:sa_mlp_max"):
                res.setdefault(k, {})[rk] = sum(a.elapsed_time(b) for a, b in v) / 3
    m.encoder.record = []
    m.encode(x)
    rec, m.encoder.record = m.encoder.record, None
for k, v in res.items():
    print("%-44s off %.3f ms   K<=4 %.3f   K<=8 %.3f" % (k, v["-1"], v["4"], v["0"]))
for l in range(2):
    for s_ in range(2):
        bi = rec[l]["ball_idx"][s_]
        K = 1 + (bi[:, :, 1:] != bi[:, :, :1]).sum(dim=2)
        n = K.numel()
        print("level %d scale %d (ns %d): %d neighbourhoods, K = 1: %.1f %%, 2..4: %.1f %%, 5..8: %.1f %%, > 8: %.1f %%"
              % (l, s_, bi.shape[2], n, 100.0 * float((K == 1).sum()) / n, 100.0 * float(((K >= 2) & (K <= 4)).sum()) / n,
                 100.0 * float(((K >= 5) & (K <= 8)).sum()) / n, 100.0 * float((K > 8).sum()) / n))
