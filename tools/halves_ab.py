#!/usr/bin/env python
"""Experiment: the headline batch as TWO HALVES on two streams, the second half's encoder queued to start when the first half's flow is
launched -- so that the encoder's latency-bound stages (index chain, set abstraction) and the flow's matrix work share the chip -- against
the whole batch in one call.  Three schedules, alternating in one process on one box, outputs compared bit for bit (sequences are
independent: sharding is bitwise invariant):
  whole      : reconstruct(x[0:16])                                           (what bench.py times)
  halves     : reconstruct(x[0:8]) on stream A; reconstruct(x[8:16]) on stream B behind A's flow launch; synchronise every step
  pipelined  : the same without the per-step synchronisation (K steps back to back: step i + 1's first encoder runs under step i's
               second flow) -- throughput only, every step's result still complete inside the timed window
Two model instances with the same weights (a model keeps per-call state: side streams, the deferred T-NOCS handle).   (GPU)"""
import copy, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from caspr_amd import ops
from caspr_amd.models import CaSPR
from caspr_amd.utils.synthetic import seeded_state_dict, car_sequences

dev = torch.device("cuda:0")
mA = CaSPR()
mA.load_state_dict(seeded_state_dict(mA.state_dict(), 0))
mA = mA.to(dev).eval()
mB = copy.deepcopy(mA)
x, sp = car_sequences(16, 10, 2048, seed=1234)
x, ts = x.to(dev), sp[0, :, 0, 3].to(dev)
torch.manual_seed(2)
yb = torch.randn(16, 10, 2048, 3).to(dev)
SA, SB = torch.cuda.Stream(), torch.cuda.Stream()
slot = {"ev": None}
_orig = ops.cnf_rk4


def _cnf(*a, **k):
    if slot["ev"] is not None:          # the moment the first half's flow is queued
        slot["ev"].record()
        slot["ev"] = None
    return _orig(*a, **k)


ops.cnf_rk4 = _cnf


def whole(y=None):
    return mA.reconstruct(x, num_points=2048, timestamps=ts, y=y)


def halves(y=None):
    ev = torch.cuda.Event()
    slot["ev"] = ev
    with torch.cuda.stream(SA):
        oa = mA.reconstruct(x[:8], num_points=2048, timestamps=ts, y=None if y is None else y[:8])
    with torch.cuda.stream(SB):
        SB.wait_event(ev)
        ob = mB.reconstruct(x[8:], num_points=2048, timestamps=ts, y=None if y is None else y[8:])
    return oa, ob


def timed(fn, k=10, sync_each=True):
    with torch.no_grad():
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(k):
            fn()
            if sync_each:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k * 1e3


with torch.no_grad():
    SA.wait_stream(torch.cuda.current_stream()); SB.wait_stream(torch.cuda.current_stream())
    ref = whole(yb)
    oa, ob = halves(yb)
    torch.cuda.synchronize()
same = torch.equal(torch.cat([oa[2], ob[2]]), ref[2]) and torch.equal(torch.cat([oa[3], ob[3]]), ref[3])
print("halves vs whole batch: outputs %s" % ("identical" if same else "DIFFER"), flush=True)
for rnd in range(3):
    a = timed(whole, sync_each=True)
    b = timed(halves, sync_each=True)
    c = timed(halves, sync_each=False)
    d = timed(whole, sync_each=False)
    print("round %d  whole %.2f ms | halves, synchronised every step %.2f ms | halves, %d steps back to back %.2f ms | whole, back to back %.2f ms"
          % (rnd, a, b, 10, c, d), flush=True)
