#!/usr/bin/env python
"""Do the weight-gradient product and the data-gradient conv of one CNF hidden layer (both consume the same dZ) run faster side by side on
two streams than one after the other?  cfg-3 training shape: 163,840 value / tangent rows, 512 x 512."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from caspr_amd import ops, lib as _lib, train_ops as T
from caspr_amd.ops import _p, _workspace
dev = torch.device("cuda:0")
frames, n, C = 80, 1024, 512
R2 = frames * 2 * n
g = torch.Generator(device="cpu").manual_seed(0)
dz = torch.randn(R2, C, generator=g).to(dev)
x = torch.randn(R2, C, generator=g).to(dev)
z = torch.randn(R2, C, generator=g).to(dev)
w = (torch.randn(C, C, generator=g) / C ** 0.5).to(dev)
b = torch.randn(C, generator=g).to(dev) * 0.1
gate = torch.sigmoid(torch.randn(frames, C, generator=g)).to(dev)
beta = (torch.randn(frames, C, generator=g) * 0.3).to(dev)
pwt = ops.PackedWeight(w.t().contiguous())
L = _lib.load()
out = torch.empty(R2, C, device=dev)
dg, db = torch.empty(frames, C, device=dev), torch.empty(frames, C, device=dev)
ws = _workspace(L.caspr_conv1x1_cnf_act_bwd_ws_bytes(frames, n, C), dev)
dw = torch.empty(C, C, device=dev)
wws = torch.empty(L.caspr_wgrad_ws_bytes(R2, C, C), device=dev, dtype=torch.uint8)
s2 = torch.cuda.Stream()

def wgrad(stream):
    _lib.check(L.caspr_conv1x1_wgrad_bf16x6_f32(_p(dz), C, _p(x), C, None, None, 0, 0, 1, R2, C, C, _p(dw), None, 0, _p(wws), wws.numel(), stream), "wgrad")

def dgrad_act(stream):
    _lib.check(L.caspr_conv1x1_cnf_act_bwd_bf16x6_f32(_p(pwt.x3()), _p(dz), C, _p(z), C, _p(b), _p(gate), _p(beta), _p(out), C, _p(dg), _p(db), _p(ws), ws.numel(),
                                                      frames, n, C, C, stream), "dgrad")

def dgrad_plain(stream):
    main, tail = pwt.xw()
    _lib.check(L.caspr_conv1x1_x6w_f32(_p(main), _p(tail), None, None, _p(dz), C, None, None, 0, 0, _p(out), C, 1, R2, C, C, 0, None, None, 0.0, None, None, None,
                                       None, None, None, 0, stream), "x6w")

def timeit(fn, reps=20):
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

main = torch.cuda.current_stream()
def serial(dgrad):
    wgrad(main.cuda_stream)
    dgrad(main.cuda_stream)
def side_by_side(dgrad, first):
    ev = torch.cuda.Event()
    ev.record(main)
    s2.wait_event(ev)
    if first == "wgrad":
        wgrad(s2.cuda_stream)
        dgrad(main.cuda_stream)
    else:
        dgrad(main.cuda_stream)
        wgrad(s2.cuda_stream)
    ev2 = torch.cuda.Event()
    ev2.record(s2)
    main.wait_event(ev2)

print("weight gradient alone                         %.3f ms" % timeit(lambda: wgrad(main.cuda_stream)))
for name, dgrad in (("dgrad + activation backward (tile kernel)", dgrad_act), ("plain dgrad (persistent kernel)", dgrad_plain)):
    print("%-45s %.3f ms" % (name + " alone", timeit(lambda: dgrad(main.cuda_stream))))
    print("  one after the other                         %.3f ms" % timeit(lambda: serial(dgrad)))
    print("  side by side, weight gradient issued first  %.3f ms" % timeit(lambda: side_by_side(dgrad, "wgrad")))
    print("  side by side, data gradient issued first    %.3f ms" % timeit(lambda: side_by_side(dgrad, "dgrad")))
