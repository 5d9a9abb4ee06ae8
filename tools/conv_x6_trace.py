#!/usr/bin/env python
"""Per-chunk cycle trace of conv1x1_bf16x6_kernel (workgroup 0, thread 0) via s_memtime stamps, plus timings of the path's
main conv shapes.  Needs the debug flavour: CASPR_BUILD_DEBUG=1 python caspr_amd/csrc/build.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from caspr_amd import lib
lib.SO_PATH = lib.SO_PATH.replace("libcaspr_hip.so", "libcaspr_hip_debug.so")
from caspr_amd import ops
dev = torch.device("cuda:0")
so = ctypes.CDLL(lib.SO_PATH)
shapes = [(1, 2560, 1600, 1600, True), (16, 20480, 1600, 1600, True), (16, 20480, 576, 1600, True), (160, 2048, 512, 512, True), (160, 2048, 544, 512, False)]
for sb in [0]:
  for (B, P, cin, cout, fused) in shapes:
      x = torch.randn(B, P, cin, device=dev)
      w = torch.randn(cout, cin, device=dev) * 0.05
      pw = ops.PackedWeight(w)
      bias = torch.randn(cout, device=dev)
      sc = torch.rand(B, cin, device=dev) + 0.5 if fused else None
      sh = torch.randn(B, cin, device=dev) if fused else None
      out = torch.empty(B, P, cout, device=dev)
      run = lambda: ops.conv1x1(pw, bias, x, in_scale=sc, in_shift=sh, in_relu=fused, out=out)
      run(); run(); torch.cuda.synchronize()
      a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      a.record()
      for _ in range(5): run()
      b.record(); torch.cuda.synchronize()
      ms = a.elapsed_time(b) / 5
      print("B=%4d P=%7d Cin=%5d Cout=%5d fused=%d  %8.3f ms  %7.1f TFLOP/s" % (B, P, cin, cout, fused, ms, 2.0 * B * P * cin * cout / ms / 1e9))
      buf = torch.zeros(160, dtype=torch.int64, device=dev)
      so.caspr_debug_set_conv_x6_trace(ctypes.c_void_p(buf.data_ptr()))
      run(); torch.cuda.synchronize()
      so.caspr_debug_set_conv_x6_trace(ctypes.c_void_p(0))
      t = buf.cpu().tolist()
      nk = min(cin // 32, 32)
      rows = []
      for kc in range(1, nk - 1):
          s = t[5 * kc: 5 * kc + 5] + [t[5 * kc + 5]]
          rows.append([s[1] - s[0], s[2] - s[1], s[3] - s[2], s[4] - s[3], s[5] - s[0]])
      mean = [sum(r[i] for r in rows) / len(rows) for i in range(5)]
      print("   per chunk (s_memtime ticks, mean of %d chunks): products %.0f | barrier-1 wait %.0f | DMA issue + split + stores %.0f | barrier-2 wait %.0f | chunk total %.0f"
            % (len(rows), mean[0], mean[1], mean[2], mean[3], mean[4]))
      print("   chunks:", " ".join("%d/%d/%d/%d" % (r[0], r[1], r[2], r[3]) for r in rows[:10]))
