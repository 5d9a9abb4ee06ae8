"""Latent ODE: 32-workgroup team kernel vs the single-workgroup kernel (difference and time).  usage: PYTHONPATH=. python tools/latent_team_bench.py"""
import time, torch
from caspr_amd import ops
from caspr_amd.models import CaSPR
from caspr_amd.utils.synthetic import seeded_state_dict
dev = torch.device("cuda:0")
m = CaSPR(); m.load_state_dict(seeded_state_dict(m.state_dict(), 0)); m = m.to(dev).eval()
lat = m.latent_ode
for B, Tu in ((16, 10), (1, 10), (5, 4), (40, 20)):
    z0 = torch.randn(B, 64, device=dev) * 0.5
    t = torch.linspace(0, 1, Tu, device=dev)
    ops.LATENT_TEAM = False
    a = ops.latent_rk4(z0, t, 2, lat._weights())
    ops.LATENT_TEAM = True
    b = ops.latent_rk4(z0, t, 2, lat._weights())
    torch.cuda.synchronize()
    print("B", B, "Tu", Tu, "max diff", float((a - b).abs().max()), "max", float(a.abs().max()))
    for flag in (False, True):
        ops.LATENT_TEAM = flag
        for _ in range(3): ops.latent_rk4(z0, t, 2, lat._weights())
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): ops.latent_rk4(z0, t, 2, lat._weights())
        torch.cuda.synchronize(); print("   team" if flag else "   single", "%.3f ms" % ((time.perf_counter() - t0) * 100))
