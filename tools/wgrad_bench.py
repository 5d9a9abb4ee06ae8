"""Time caspr_conv1x1_wgrad_f32 on the training step's shapes.  usage: PYTHONPATH=. python tools/wgrad_bench.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from caspr_amd import ops, train_ops as T

dev = "cuda:0"
for mode in ("bf16x6", "f32"):
  ops.set_matmul_mode(mode)
  print("== matrix products:", mode)
  for (R, cin, cout) in [(81920, 1600, 1600), (-81920, 1600, 1600), (-163840, 512, 512), (163840, 512, 512), (81920, 576, 1600), (2621440, 12, 32), (655360, 64, 128), (163840, 3, 512)]:
      fused = R < 0
      R = abs(R)
      cinp = (cin + 3) // 4 * 4
      nb = 16 if fused else 1
      x = torch.randn(nb, R // nb, cinp, device=dev)
      dy = torch.randn(nb, R // nb, (cout + 3) // 4 * 4, device=dev)
      sc = torch.rand(nb, cin, device=dev) + 0.5 if fused else None
      sh = torch.randn(nb, cin, device=dev) if fused else None
      dw = torch.empty(cout, cin, device=dev)
      db = torch.empty(cout, device=dev)
      for _ in range(2):
          T.conv1x1_wgrad(dy, x, cin, cout, dw, db, in_scale=sc, in_shift=sh, in_relu=fused)
      torch.cuda.synchronize()
      a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      a.record()
      n = 5
      for _ in range(n):
          T.conv1x1_wgrad(dy, x, cin, cout, dw, db, in_scale=sc, in_shift=sh, in_relu=fused)
      b.record()
      torch.cuda.synchronize()
      ms = a.elapsed_time(b) / n
      print("R=%8d Cin=%5d Cout=%5d fused=%d  %8.3f ms  %7.1f TFLOP/s" % (R, cin, cout, fused, ms, 2.0 * R * cin * cout / ms / 1e9))
