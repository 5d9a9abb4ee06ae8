#!/usr/bin/env python
"""Host-side clock of one reconstruct() call at cfg-2 (no synchronisation inside): when is each part ENQUEUED, against the GPU's
stage times?  If the host reaches the latent solve later than the GPU finishes the encoder, the GPU waits for Python."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from caspr_amd.models import CaSPR
from caspr_amd.utils.synthetic import seeded_state_dict, car_sequences
dev = torch.device("cuda:0")
m = CaSPR(); m.load_state_dict(seeded_state_dict(m.state_dict(), 0)); m = m.to(dev).eval()
x, _ = car_sequences(16, 10, 2048)
x = x.to(dev)
ts = torch.linspace(0, 1, 10)
for _ in range(3):
    m.reconstruct(x, num_points=2048, timestamps=ts)
torch.cuda.synchronize()
marks = []
orig_enc, orig_draw, orig_lat, orig_dec = m.encoder.forward, m._draw_early, m.aggregate_and_solve_latent, m.decode
def wrap(name, fn):
    def f(*a, **k):
        t0 = time.perf_counter(); r = fn(*a, **k); marks.append((name, t0, time.perf_counter())); return r
    return f
m.encoder.forward = wrap("encoder enqueue", orig_enc)
m._draw_early = wrap("base-sample draw", orig_draw)
m.aggregate_and_solve_latent = wrap("latent enqueue", orig_lat)
m.decode = wrap("decode enqueue", orig_dec)
for rep in range(3):
    marks.clear()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    m.reconstruct(x, num_points=2048, timestamps=ts)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("call returned after %.2f ms, GPU done after %.2f ms: " % ((t1 - t0) * 1e3, (t2 - t0) * 1e3) +
          ", ".join("%s %.2f-%.2f" % (n, (a - t0) * 1e3, (b - t0) * 1e3) for n, a, b in marks))
