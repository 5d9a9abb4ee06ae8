#!/usr/bin/env python
"""A/B timing of the bf16x6 pointwise conv kernels on the model's large layers (debug flavour of the library:
CASPR_BUILD_DEBUG=1 python caspr_amd/csrc/build.py).  CASPR_X6_CONV_SINGLE=1 selects the single-buffered kernel."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from caspr_amd import lib
# experiment switches live in the debug flavour only: CASPR_BUILD_DEBUG=1 python caspr_amd/csrc/build.py
lib.SO_PATH = lib.SO_PATH.replace("libcaspr_hip.so", "libcaspr_hip_debug.so")
from caspr_amd import ops
dev = torch.device("cuda:0")
shapes = [("head conv2", 16, 20480, 1600, 1600, True), ("head conv1", 16, 20480, 576, 1600, True), ("FP 512->512", 160, 2048, 512, 512, True),
          ("FP 768->512", 160, 1024, 768, 512, False), ("cfg-5 head", 8, 81920, 1600, 1600, True)]
for name, B, P, Cin, Cout, fused in shapes:
    x = torch.randn(B, P, Cin, device=dev)
    w = torch.randn(Cout, Cin, device=dev) / Cin ** 0.5
    pw = ops.PackedWeight(w)
    sc = torch.rand(B, Cin, device=dev) + 0.5 if fused else None
    sh = torch.randn(B, Cin, device=dev) if fused else None
    out = torch.empty(B, P, Cout, device=dev)
    res = {}
    for mode, env in (("double", "0"), ("single", "1"), ("f32", None)):
        if env is None:
            ops.set_matmul_mode("f32")
        else:
            ops.set_matmul_mode("bf16x6")
            os.environ["CASPR_X6_CONV_SINGLE"] = env
        for _ in range(2):
            ops.conv1x1(pw, None, x, in_scale=sc, in_shift=sh, in_relu=fused, out=out)
        torch.cuda.synchronize()
        t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(5):
            ops.conv1x1(pw, None, x, in_scale=sc, in_shift=sh, in_relu=fused, out=out)
        t1.record(); torch.cuda.synchronize()
        ms = t0.elapsed_time(t1) / 5
        res[mode] = (ms, out.clone())
    fl = 2.0 * B * P * Cin * Cout
    d = float((res["double"][1] - res["single"][1]).abs().max())
    print("%-12s B=%d P=%d %d->%d: double %.3f ms (%.0f TF)  single %.3f ms (%.0f TF)  f32 %.3f ms (%.0f TF)  |double-single| %.1e" % (
        name, B, P, Cin, Cout, res["double"][0], fl / res["double"][0] / 1e9, res["single"][0], fl / res["single"][0] / 1e9,
        res["f32"][0], fl / res["f32"][0] / 1e9, d))
