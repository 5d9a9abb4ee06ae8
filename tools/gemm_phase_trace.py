#!/usr/bin/env python
"""Per-K-tile cycle breakdown of conv1x1_kernel (one wave of one block, K tiles 8..11) on the head's conv2 shape."""
import ctypes, sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from caspr_amd import lib, ops
# phase traces / experiment switches live in the debug flavour only: CASPR_BUILD_DEBUG=1 python caspr_amd/csrc/build.py
lib.SO_PATH = lib.SO_PATH.replace("libcaspr_hip.so", "libcaspr_hip_debug.so")
dev = torch.device("cuda:0")
so = ctypes.CDLL(lib.SO_PATH)
B, P, Cin, Cout = 16, 20480, 1600, 1600
x = torch.randn(B, P, Cin, device=dev)
pw = ops.PackedWeight(torch.randn(Cout, Cin, device=dev) * 0.02)
bias = torch.zeros(Cout, device=dev)
sc, sh = torch.ones(B, Cin, device=dev), torch.zeros(B, Cin, device=dev)
y = ops.conv1x1(pw, bias, x, in_scale=sc, in_shift=sh, in_relu=True); torch.cuda.synchronize()
buf = torch.zeros(64, dtype=torch.int64, device=dev)
so.caspr_debug_set_gemm_trace(ctypes.c_void_p(buf.data_ptr()))
t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True); t0.record()
y = ops.conv1x1(pw, bias, x, in_scale=sc, in_shift=sh, in_relu=True, out=y)
t1.record(); torch.cuda.synchronize()
so.caspr_debug_set_gemm_trace(ctypes.c_void_p(0))
ms = t0.elapsed_time(t1)
print("conv2 shape: %.3f ms, %.1f TFLOP/s" % (ms, 2.0 * B * P * Cin * Cout / ms / 1e9))
t = buf.cpu().view(8, 8)
names = ["issue loads", "mma chunk0 (64)", "mma chunk1 (64)", "store_stage", "barrier"]
for it in range(4):
    d = (t[it, 1:6] - t[it, 0:5]).tolist()
    print("K tile %d: total %d cycles: " % (it + 8, int(t[it, 5] - t[it, 0])) + ", ".join("%s %d" % (n, v) for n, v in zip(names, d)))
