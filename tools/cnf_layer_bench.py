#!/usr/bin/env python
"""The CNF's 512 -> 512 hidden layer at the training shape (80 frames x 2 x 1024 value / tangent rows): the plain conv, the conv with the
gated softplus in its read-out (forward) and the data-gradient conv with the activation's backward.  (Round 4 also had both
activation read-outs inside the persistent 512-channel kernel: DESIGN.md section 3, "two output streams", has the numbers and why it was dropped.)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from caspr_amd import ops, lib as _lib
from caspr_amd.ops import _p, _stream, _workspace
dev = torch.device("cuda:0")
frames, n, C = 80, 1024, 512
R2 = frames * 2 * n
g = torch.Generator(device="cpu").manual_seed(0)
x = torch.randn(R2, C, generator=g).to(dev)
w = (torch.randn(C, C, generator=g) / C ** 0.5).to(dev)
b = torch.randn(C, generator=g).to(dev) * 0.1
gate = torch.sigmoid(torch.randn(frames, C, generator=g)).to(dev)
beta = (torch.randn(frames, C, generator=g) * 0.3).to(dev)
pw = ops.PackedWeight(w)
L = _lib.load()
z, h = torch.empty(R2, C, device=dev), torch.empty(R2, C, device=dev)
dg, db = torch.empty(frames, C, device=dev), torch.empty(frames, C, device=dev)
ws = _workspace(L.caspr_conv1x1_cnf_act_bwd_ws_bytes(frames, n, C), dev)
xv = x.view(frames, 2 * n, C)

def t(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return a.elapsed_time(e) / reps

def plain(x6w):
    old = ops.CONV_X6W
    ops.CONV_X6W = x6w
    try:
        return t(lambda: ops.conv1x1(pw, b, xv, out=h.view(frames, 2 * n, C)))
    finally:
        ops.CONV_X6W = old

fl = 2.0 * R2 * C * C / 1e9
rows = [("plain conv, 256-channel tile kernel", plain(False)), ("plain conv, persistent 512-channel kernel", plain(True)),
        ("conv + activation (fwd), 256-channel tile kernel", t(lambda: _lib.check(L.caspr_conv1x1_cnf_act_bf16x6_f32(_p(pw.x3()), _p(b), _p(gate), _p(beta), _p(x), C, _p(z), C, _p(h), C, frames, n, C, C, _stream()), "a"))),
        ("dgrad conv + activation backward, 256-channel tile kernel", t(lambda: _lib.check(L.caspr_conv1x1_cnf_act_bwd_bf16x6_f32(_p(pw.x3()), _p(x), C, _p(z), C, _p(b), _p(gate), _p(beta), _p(h), C, _p(dg), _p(db), _p(ws), ws.numel(), frames, n, C, C, _stream()), "c")))]
for name, ms in rows:
    print("%-62s %7.3f ms  %6.1f TFLOP/s (f32-eq)" % (name, ms, fl / ms))
