#!/usr/bin/env python
"""encode() at the headline shape, many runs, NO instrumentation inside the pipeline: how often do z0 / T-NOCS differ from the first run, and where
(sequence, channel tile)?  Optional interleaving with reconstruct() calls of other shapes (allocator / stream history, as a test suite has)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from caspr_amd.models import CaSPR
from caspr_amd.utils.synthetic import seeded_state_dict, car_sequences
dev = torch.device("cuda:0")
RUNS = int(sys.argv[1]) if len(sys.argv) > 1 else 300
MIX = "--mix" in sys.argv
x, sp = car_sequences(16, 10, 2048, seed=1234)
m = CaSPR(); m.load_state_dict(seeded_state_dict(m.state_dict(), 0)); m = m.to(dev).eval()
xg = x.to(dev)
x2, sp2 = car_sequences(3, 4, 1024, seed=5)
x2g, ts2 = x2.to(dev), sp2[0, :, 0, 3].to(dev)
base, nbad = None, 0
for r in range(RUNS):
    if MIX and r % 3 == 1:
        with torch.no_grad():
            m.reconstruct(x2g, num_points=256, timestamps=ts2)
    with torch.no_grad():
        z0, tn = m.encode(xg)
    torch.cuda.synchronize()
    if base is None:
        base = (z0.clone(), tn.clone()); continue
    dz, dt = (z0 != base[0]), (tn != base[1])
    if dz.any() or dt.any():
        nbad += 1
        seqs = dz.any(dim=1).nonzero().flatten().tolist()
        tiles = {s: [int(dz[s, 512 * t_:512 * (t_ + 1)].sum()) for t_ in range(4)] for s in seqs}
        print("run %d: z0 differs in %d entries (per sequence: 512-channel tiles %s), max |dz| %.3e; tnocs differs in %d entries of sequences %s" % (
            r, int(dz.sum()), tiles, float((z0 - base[0]).abs().max()), int(dt.sum()), dt.flatten(1).any(dim=1).nonzero().flatten().tolist()), flush=True)
print("%d of %d runs differ from run 0 (mix=%s)" % (nbad, RUNS - 1, MIX))
