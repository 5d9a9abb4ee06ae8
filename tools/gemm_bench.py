"""Time caspr_conv1x1_f32 on the path's main shapes.  usage: PYTHONPATH=. python tools/gemm_bench.py"""
import torch
from caspr_amd import ops

dev = "cuda:0"
for (B, P, cin, cout, fused) in [(16, 20480, 1600, 1600, True), (16, 20480, 576, 1600, True), (1, 163840, 512, 512, False), (160, 2048, 512, 512, True),
                                 (160, 1024, 608, 512, True), (1, 163840, 512, 512, True), (160, 1024, 512, 512, False), (80, 2048, 512, 512, False), (16, 20480, 1600, 1600, False)]:
    x = torch.randn(B, P, cin, device=dev)
    w = torch.randn(cout, cin, device=dev) * 0.05
    pw = ops.PackedWeight(w)
    bias = torch.randn(cout, device=dev)
    sc = torch.rand(B, cin, device=dev) + 0.5 if fused else None
    sh = torch.randn(B, cin, device=dev) if fused else None
    out = torch.empty(B, P, cout, device=dev)
    for _ in range(2):
        ops.conv1x1(pw, bias, x, in_scale=sc, in_shift=sh, in_relu=fused, out=out)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 5
    a.record()
    for _ in range(n):
        ops.conv1x1(pw, bias, x, in_scale=sc, in_shift=sh, in_relu=fused, out=out)
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / n
    print("B=%4d P=%7d Cin=%5d Cout=%5d fused=%d  %8.3f ms  %7.1f TFLOP/s" % (B, P, cin, cout, fused, ms, 2.0 * B * P * cin * cout / ms / 1e9))
