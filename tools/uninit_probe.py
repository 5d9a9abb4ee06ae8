#!/usr/bin/env python
"""Does any kernel of the training step (or of reconstruct()) read memory nobody wrote?  The step once on a clean allocator, then the
caching allocator's free blocks are filled with a poison value and the step runs again: every output must keep its bits.   (GPU)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from caspr_amd.models import CaSPR
from caspr_amd.utils.synthetic import seeded_state_dict, car_sequences
from caspr_amd.train.loop import training_loss

dev = torch.device("cuda:0")
B, T, N = 2, 2, 1024
x, sp = car_sequences(B, T, N, seed=11)
torch.manual_seed(3)
e = torch.randn(B * T, N, 3)


def poison(value):
    torch.cuda.synchronize()
    blocks = []
    try:
        for _ in range(24):
            blocks.append(torch.full((1 << 30,), value, device=dev, dtype=torch.float32))     # 4 GB each
    except RuntimeError:
        pass
    del blocks                      # back to the caching allocator, contents intact
    torch.cuda.synchronize()


def train_grads():
    m = CaSPR(cnf_rk4_steps=4, latent_rk4_steps=2)
    m.load_state_dict(seeded_state_dict(m.state_dict(), 0))
    m = m.to(dev).train()
    loss, _, _ = training_loss(m(x.to(dev), sp.to(dev), e=e.to(dev)), 0.01, 100.0)
    loss.backward()
    torch.cuda.synchronize()
    return float(loss), {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}


def recon():
    m = CaSPR()
    m.load_state_dict(seeded_state_dict(m.state_dict(), 0))
    m = m.to(dev).eval()
    torch.manual_seed(1)
    with torch.no_grad():
        o = m.reconstruct(x.to(dev), num_points=256, timestamps=sp[0, :, 0, 3].to(dev))
    torch.cuda.synchronize()
    return o[2].clone(), o[3].clone()


for name, fn in (("reconstruct", recon), ("training step", train_grads)):
    ref = fn()
    for val in (float("nan"), 1e30, -7.0):
        poison(val)
        got = fn()
        if name == "reconstruct":
            bad = [i for i, (a, b_) in enumerate(zip(ref, got)) if not torch.equal(a, b_)]
            print("%s, free memory poisoned with %s: %s" % (name, val, "identical bits" if not bad else "DIFFERS in outputs %s" % bad), flush=True)
        else:
            bad = [n for n in ref[1] if not torch.equal(ref[1][n], got[1][n])]
            nan = [n for n in bad if not torch.isfinite(got[1][n]).all()]
            print("%s, free memory poisoned with %s: loss %r vs %r; %d of %d gradient tensors differ (%d non-finite)%s"
                  % (name, val, ref[0], got[0], len(bad), len(ref[1]), len(nan), (": " + ", ".join(bad[:8])) if bad else ""), flush=True)
