import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from caspr_amd.models.latent_ode_model import LatentODE
from caspr_amd.train import flow_grad as FG
dev = torch.device("cuda:0")
lat = LatentODE(input_size=64, hidden_size=512, num_layers=2).to(dev)
lat.rk4_steps = 2
z0 = torch.randn(8, 64, device=dev, requires_grad=True)
times = torch.linspace(0, 1, 10, device=dev)
w = torch.randn(8, 10, 64, device=dev)
def run():
    out = FG.latent_solve_train(lat, z0, times)
    (out * w).sum().backward()
for _ in range(3): run()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): run()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("latent forward + backward (8 x 10, 72 evaluations): host %.2f ms per call, with the GPU %.2f ms per call" % ((t1 - t0) * 100, (t2 - t0) * 100))
