import sys, os
sys.path.insert(0, "/root/repo")
import torch
from caspr_amd.models import CaSPR
from caspr_amd.utils.synthetic import seeded_state_dict, car_sequences
dev = torch.device("cuda:0")
m = CaSPR(); m.load_state_dict(seeded_state_dict(m.state_dict(), 0)); m = m.to(dev).eval()
x, _ = car_sequences(16, 10, 2048, seed=1234); x = x.to(dev)
with torch.no_grad():
    for _ in range(6):
        m.encode(x)
torch.cuda.synchronize()
