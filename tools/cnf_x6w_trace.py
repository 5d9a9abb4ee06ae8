#!/usr/bin/env python
"""Per-piece cycle trace of cnf_rk4_x6w_kernel (workgroup (0,0), thread 0, RK4 step 0, stage 1) via s_memtime stamps, and
an A/B timing against the 64-point kernel.  Needs the debug flavour: CASPR_BUILD_DEBUG=1 python caspr_amd/csrc/build.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from caspr_amd import lib
lib.SO_PATH = lib.SO_PATH.replace("libcaspr_hip.so", "libcaspr_hip_debug.so")
from caspr_amd.models import CaSPR
from caspr_amd.utils.synthetic import seeded_state_dict
dev = torch.device("cuda:0")
m = CaSPR(); m.load_state_dict(seeded_state_dict(m.state_dict(), 0)); m = m.to(dev).eval()
BT, n = 160, 2048
y, c = torch.randn(BT, n, 3, device=dev), torch.randn(BT, 1600, device=dev)
so = ctypes.CDLL(lib.SO_PATH)
def run(k=3):
    with torch.no_grad():
        m.point_cnf(y, c, reverse=True); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(k): m.point_cnf(y, c, reverse=True)
        b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / k
for rep in range(2):
    os.environ["CASPR_X6_WIDE"] = "1"; tw = run()
    os.environ["CASPR_X6_WIDE"] = "0"; tn = run()
    print("wide %.2f ms   narrow %.2f ms (includes the hyper conv)" % (tw, tn))
os.environ["CASPR_X6_WIDE"] = "1"
buf = torch.zeros(288, dtype=torch.int64, device=dev)
so.caspr_debug_set_x6_trace(ctypes.c_void_p(buf.data_ptr()))
with torch.no_grad():
    m.point_cnf(y, c, reverse=True)
torch.cuda.synchronize()
so.caspr_debug_set_x6_trace(ctypes.c_void_p(0))
t = buf.cpu().tolist()
ph = lambda s_: (t[2 * s_], t[2 * s_ + 1])
print("stage total (stage 1 start -> stage 2 start): %d cycles" % (t[272] - t[271]))
print("prologue (tables, chunk 0) until piece 0 arrives at its barrier: %d" % (t[0] - t[271]))
def piece_stats(rng, label):
    waits = [t[2 * s_ + 1] - t[2 * s_] for s_ in rng]
    spans = [t[2 * (s_ + 1)] - t[2 * s_ + 1] for s_ in rng if s_ + 1 in rng or s_ + 1 < 128]
    print("%-28s pieces %3d: body mean %6.0f (min %d max %d)  barrier wait mean %5.0f (max %d)" % (
        label, len(list(rng)), sum(spans) / max(len(spans), 1), min(spans), max(spans), sum(waits) / len(waits), max(waits)))
piece_stats(range(0, 63), "layer 1")
print("  layer-1 bodies by region kind: tab pieces %s" % [t[2 * (s_ + 1)] - t[2 * s_ + 1] for s_ in range(8, 16)])
print("layer 1 flush + pass 0 chunk-0 producers: %d" % (t[2 * 64] - t[2 * 63 + 1]))
for q in range(4):
    piece_stats(range(64 + 16 * q, 64 + 16 * q + 15), "layer 2 pass %d" % q)
    print("  pass %d: last piece + flush %d, epilogue %d" % (q, t[257 + 2 * q] - t[2 * (64 + 16 * q + 15) + 1], t[258 + 2 * q] - t[257 + 2 * q]))
print("end of stage (output layer, update): %d" % (t[272] - t[258 + 6]))
