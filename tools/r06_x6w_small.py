import os, sys
sys.path.insert(0, '/root/repo')
import torch
from caspr_amd import ops
dev = torch.device("cuda:0")
def t(fn, k=10):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(k): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / k
for (B, P, Cin, Cout, write) in [(16, 20480, 128, 1024, False), (160, 512, 256, 512, True), (160, 256, 256, 512, True), (160, 256, 768, 512, True), (160, 256, 512, 512, True), (160, 512, 640, 512, True), (160, 512, 512, 512, True)]:
    w = torch.randn(Cout, Cin, device=dev) / Cin ** 0.5
    bias = torch.randn(Cout, device=dev); x = torch.randn(B, P, Cin, device=dev)
    sc, sh = torch.rand(B, Cin, device=dev) + 0.5, torch.randn(B, Cin, device=dev)
    g, be = torch.ones(Cout, device=dev), torch.zeros(Cout, device=dev)
    pw = ops.PackedWeight(w)
    res = []
    for mc in (512, 128):
        ops._X6W_MIN_CIN = mc
        ms = t(lambda: ops.conv1x1_gn(pw, bias, x, g, be, in_scale=sc, in_shift=sh, in_relu=True, want_max=True, write=write))
        res.append("min_cin %d: %.3f ms (%.0f TF)" % (mc, ms, 2.0 * B * P * Cin * Cout / ms / 1e9))
    print("B=%d P=%d %d->%d write=%s: %s" % (B, P, Cin, Cout, write, "   ".join(res)), flush=True)
