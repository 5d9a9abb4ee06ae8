"""reconstruct() per-call latency, eager vs hipGraph replay, at small batch.  usage: PYTHONPATH=. python tools/graph_latency.py"""
import time, torch
from caspr_amd.models import CaSPR
from caspr_amd.utils.synthetic import seeded_state_dict, car_sequences
from caspr_amd.utils.graphs import GraphedReconstruct
dev = torch.device("cuda:0")
m = CaSPR(); m.load_state_dict(seeded_state_dict(m.state_dict(), 0)); m = m.to(dev).eval()
for B in (1, 4):
    x, sp = car_sequences(B, 10, 2048, seed=5)
    x = x.to(dev); ts = sp[0, :, 0, 3].to(dev)
    y = torch.randn(B, 10, 2048, 3, device=dev)
    ref = m.reconstruct(x, num_points=2048, timestamps=ts, y=y)
    g = GraphedReconstruct(m, x, 2048, ts)
    out = g(x, y)
    torch.cuda.synchronize()
    print("B", B, "identical:", all(torch.equal(a, b) for a, b in zip(ref, out)))
    for name, fn in (("eager", lambda: m.reconstruct(x, num_points=2048, timestamps=ts, y=y)), ("graph", lambda: g(x, y))):
        for _ in range(3): fn()
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(10): fn()
        torch.cuda.synchronize(); print("  %s %.2f ms/call" % (name, (time.perf_counter() - t) * 100))
