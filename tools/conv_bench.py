#!/usr/bin/env python
"""Timing of the pointwise conv kernels on the encoder's large layers at cfg-2 sizes: the 128-point x 512-channel kernel
(csrc/gemm_bf16x6w.hip) against the 256-channel one (csrc/gemm_bf16x6.hip) and the f32-MFMA kernel."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from caspr_amd import lib
if "--lib" in sys.argv:
    lib.SO_PATH = lib.SO_PATH.replace("libcaspr_hip.so", sys.argv[sys.argv.index("--lib") + 1])
from caspr_amd import ops
if "--min-cin" in sys.argv:          # let the 512-channel kernel take layers below ops._X6W_MIN_CIN input channels
    ops._X6W_MIN_CIN = int(sys.argv[sys.argv.index("--min-cin") + 1])
dev = torch.device("cuda:0")
def t(fn, k=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(k): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / k
for (B, P, Cin, Cout, gn) in [(16, 20480, 1600, 1600, True), (16, 20480, 1600, 1600, False), (16, 20480, 576, 1600, True), (160, 2048, 544, 512, True), (160, 1024, 608, 512, True), (160, 2048, 512, 512, True), (160, 512, 640, 512, True)]:
    w = torch.randn(Cout, Cin, device=dev) / Cin ** 0.5
    bias = torch.randn(Cout, device=dev)
    x = torch.randn(B, P, Cin, device=dev)
    sc, sh = torch.rand(B, Cin, device=dev) + 0.5, torch.randn(B, Cin, device=dev)
    g, be = torch.ones(Cout, device=dev), torch.zeros(Cout, device=dev)
    pw = ops.PackedWeight(w)
    out = torch.empty(B, P, Cout, device=dev)
    def run():
        if gn: ops.conv1x1_gn(pw, bias, x, g, be, in_scale=sc, in_shift=sh, in_relu=True, out=out)
        else: ops.conv1x1(pw, bias, x, in_scale=sc, in_shift=sh, in_relu=True, out=out)
    fl = 2.0 * B * P * Cin * Cout
    res = []
    for name, x6w, mode in (("x6w", True, "bf16x6"), ("x6", False, "bf16x6"), ("f32", False, "f32")):
        ops.CONV_X6W = x6w
        prev = ops.set_matmul_mode(mode)
        ms = t(run)
        ops.set_matmul_mode(conv=prev[0], cnf=prev[1])
        res.append("%s %.3f ms %.0f TF" % (name, ms, fl / ms / 1e9))
    ops.CONV_X6W = True
    print("B=%d P=%d %d->%d gn=%d: %s" % (B, P, Cin, Cout, gn, "   ".join(res)))
