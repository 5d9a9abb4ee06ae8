#!/usr/bin/env python
"""Six cfg-2 encoder passes (car clouds, 16 x 10 x 2048) for rocprofv3 --kernel-trace --stats; optional arguments: the library's file
name; --one-stream: the two scales of a set-abstraction level one after the other (per-kernel durations without the other scale's)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from caspr_amd import lib
one_stream = "--one-stream" in sys.argv
argv = [a for a in sys.argv[1:] if not a.startswith("--")]
if argv:
    lib.SO_PATH = lib.SO_PATH.replace("libcaspr_hip.so", argv[0])
import torch
from caspr_amd.models import CaSPR
from caspr_amd.utils.synthetic import seeded_state_dict, car_sequences
if one_stream:
    import caspr_amd.models.pointnet2 as P2
    P2.SCALE_STREAMS = False
dev = torch.device("cuda:0")
m = CaSPR()
m.load_state_dict(seeded_state_dict(m.state_dict(), 0))
m = m.to(dev).eval()
x, _ = car_sequences(16, 10, 2048, seed=1234)
x = x.to(dev)
with torch.no_grad():
    for _ in range(6):
        m.encode(x)
torch.cuda.synchronize()
