"""How much does the f32 encoder gradient move under a 1-ulp-scale perturbation of the input?  (conditioning probe
for tests/test_hip_train.py: the HIP gradient is compared with f64 autograd, and this is the noise floor any f32
implementation has.)  Prints the relative L2 distance between the gradients at x and at x*(1+1e-7*u)."""
import json
import numpy as np
import torch

from caspr_amd.models import CaSPR
from caspr_amd.utils.synthetic import seeded_state_dict, dense_sequences

dev = "cuda:0"
m = CaSPR(pretrain_tnocs=True)
sd = {k: v for k, v in seeded_state_dict(CaSPR().state_dict(), seed=7).items() if k.startswith("encoder.")}
m.load_state_dict(sd)
m = m.to(dev).train()
x, sp = dense_sequences(1, 2, 1024)
R = torch.from_numpy((np.random.default_rng(3).normal(0, 1, (1, 1600)) * 0.05).astype(np.float32)).to(dev)


def grads(xx):
    m.zero_grad()
    z0, tn = m.encoder(xx.to(dev))
    (100.0 * (tn - sp.to(dev)).abs().mean() + (z0 * R).sum()).backward()
    return {n: p.grad.detach().double().clone() for n, p in m.named_parameters()}


g0 = grads(x)
g0b = grads(x)
u = torch.from_numpy(np.random.default_rng(1).uniform(-1, 1, x.shape).astype(np.float32))
xp = x.clone()
xp[..., :3] = x[..., :3] * (1 + 2e-7 * u[..., :3])
g1 = grads(xp)


def dist(a, b):
    num = sum(float((a[k] - b[k]).norm()) ** 2 for k in a)
    den = sum(float(a[k].norm()) ** 2 for k in a)
    return (num / den) ** 0.5


out = {"repeat_same_input": dist(g0, g0b), "input_perturbed_2e-7": dist(g0, g1),
       "moved_points": int((xp != x).any(-1).sum())}
print(json.dumps(out))
