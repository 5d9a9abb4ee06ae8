#!/usr/bin/env python
"""Per-phase cycle breakdown of cnf_rk4_kernel (workgroup (0,0), wave 0, first RK4 step) via s_memtime stamps."""
import ctypes, sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from caspr_amd import lib
# phase traces / experiment switches live in the debug flavour only: CASPR_BUILD_DEBUG=1 python caspr_amd/csrc/build.py
lib.SO_PATH = lib.SO_PATH.replace("libcaspr_hip.so", "libcaspr_hip_debug.so")
from caspr_amd.models import CaSPR
from caspr_amd.utils.synthetic import seeded_state_dict
dev = torch.device("cuda:0")
m = CaSPR(); m.load_state_dict(seeded_state_dict(m.state_dict(), 0)); m = m.to(dev).eval()
buf = torch.zeros(64, dtype=torch.int64, device=dev)
L = lib.load()
BT, n = 160, 2048
y, c = torch.randn(BT, n, 3, device=dev), torch.randn(BT, 1600, device=dev)
m.point_cnf(y, c, reverse=True); torch.cuda.synchronize()
L._handle  # noqa
ctypes.CDLL(lib.SO_PATH).caspr_debug_set_cnf_trace(ctypes.c_void_p(buf.data_ptr()))
m.point_cnf(y, c, reverse=True); torch.cuda.synchronize()
ctypes.CDLL(lib.SO_PATH).caspr_debug_set_cnf_trace(ctypes.c_void_p(0))
t = buf.cpu().view(4, 16)
names = ["gates+ys", "barrier", "layer0", "barrier", "mfma1", "epilogue1", "barrier", "write H", "barrier", "mfma2", "epilogue2+out", "barrier", "combine", "barrier"]
for st in range(4):
    d = (t[st, 1:15] - t[st, 0:14]).tolist()
    tot = int(t[st, 14] - t[st, 0])
    print("stage %d total %d cycles (s_memtime ticks): " % (st, tot) + ", ".join("%s %d" % (nm, v) for nm, v in zip(names, d)))
