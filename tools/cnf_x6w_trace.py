#!/usr/bin/env python
"""A/B timing of the two bf16x6 sampling kernels of the point CNF (128-point cnf_rk4_x6w_kernel vs 64-point cnf_rk4_x6_kernel)
and the phase breakdown of one stage of the 128-point kernel (workgroup (0,0), thread 0, RK4 step 0, stage 1) from its
s_memtime stamps.  Needs the debug flavour: CASPR_BUILD_DEBUG=1 python caspr_amd/csrc/build.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from caspr_amd import lib
LIBNAME = sys.argv[sys.argv.index("--lib") + 1] if "--lib" in sys.argv else "libcaspr_hip_debug.so"
lib.SO_PATH = lib.SO_PATH.replace("libcaspr_hip.so", LIBNAME)
from caspr_amd.models import CaSPR
from caspr_amd.utils.synthetic import seeded_state_dict
dev = torch.device("cuda:0")
m = CaSPR(); m.load_state_dict(seeded_state_dict(m.state_dict(), 0)); m = m.to(dev).eval()
BT, n = 160, 2048
y, c = torch.randn(BT, n, 3, device=dev), torch.randn(BT, 1600, device=dev)
so = ctypes.CDLL(lib.SO_PATH)
def run(k=3):
    with torch.no_grad():
        out = m.point_cnf(y, c, reverse=True); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(k): m.point_cnf(y, c, reverse=True)
        b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / k, out
for rep in range(2):
    os.environ["CASPR_X6_NARROW"] = "0"; tw, ow = run()
    os.environ["CASPR_X6_NARROW"] = "1"; tn, on = run()
    print("128-point kernel %.2f ms   64-point kernel %.2f ms (both include the hyper conv)   max |diff| %.3e" % (tw, tn, float((ow - on).abs().max())))
if "--timing-only" in sys.argv:
    sys.exit(0)
os.environ["CASPR_X6_NARROW"] = "0"
buf = torch.zeros(512, dtype=torch.int64, device=dev)
so.caspr_debug_set_x6_trace(ctypes.c_void_p(buf.data_ptr()))
with torch.no_grad():
    m.point_cnf(y, c, reverse=True)
torch.cuda.synchronize()
so.caspr_debug_set_x6_trace(ctypes.c_void_p(0))
t = buf.cpu().tolist()
names = [("tables + accumulator zeroing", 300, 301), ("layer 1: chunk 0 of the input layer (exposed)", 301, 302), ("layer 1 (64 pieces, floor 98304)", 302, 303),
         ("layer 2 pass 0 incl. epilogue (floor 24576)", 303, 304), ("layer 2 passes 1-3 incl. epilogues (floor 73728)", 304, 305),
         ("output layer + RK4 update", 305, 306)]
print("stage total: %d cycles (MFMA floor 196608)" % (t[306] - t[300]))
for nm, a_, b_ in names:
    print("  %-55s %8d" % (nm, t[b_] - t[a_]))

# per piece: stamp 2s = arrival at the barrier that opens piece s (placed in the last region of piece s - 1), 2s + 1 = past it
def stats(rng, label):
    waits = [t[2 * s_ + 1] - t[2 * s_] for s_ in rng]
    spans = [t[2 * (s_ + 1)] - t[2 * s_ + 1] for s_ in rng if s_ + 1 < 128 and (s_ + 1) in rng]
    print("%-22s pieces %3d: barrier-to-barrier mean %6.0f (min %d max %d; floor 1536)  wait at the barrier mean %5.0f (max %d)" % (
        label, len(list(rng)), sum(spans) / max(len(spans), 1), min(spans), max(spans), sum(waits) / len(waits), max(waits)))
stats(range(1, 64), "layer 1")
for q in range(4):
    stats(range(64 + 16 * q, 64 + 16 * q + 16), "layer 2 pass %d" % q)
    print("    pass %d: epilogue %d cycles; last barrier of the pass -> epilogue start %d" % (q, t[311 + 2 * q] - t[310 + 2 * q], t[310 + 2 * q] - t[2 * ((64 + 16 * q + 16) & 127) + 1]))
print("layer 1, spans by piece: %s" % [t[2 * (s_ + 1)] - t[2 * s_ + 1] for s_ in range(1, 40)])
print("pass 1, spans by piece: %s" % [t[2 * (s_ + 1)] - t[2 * s_ + 1] for s_ in range(80, 95)])

r = [t[320 + i] for i in range(64)]
print("layer 1, iterations 1-2, cycles per region (4 per piece; floor 384): %s" % [r[i + 1] - r[i] for i in range(63)])
