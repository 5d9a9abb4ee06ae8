#!/usr/bin/env python
"""Which neighbourhoods carry the set-abstraction error on cfg-5's input kind (i.i.d. uniform clouds)?  Levels 0 and 1, each scale in
isolation against the f64 evaluation of pointnet2.py:649-703 on the same grouped input; per number K of DISTINCT samples in the ball:
count, max error, how many above 1e-5.  Debug library: CASPR_SA_REPAIR_K = -1 / 4 / 8 (csrc/sa_mlp.hip).   (GPU)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from caspr_amd import lib
if "--prod" not in sys.argv:
    lib.SO_PATH = lib.SO_PATH.replace("libcaspr_hip.so", "libcaspr_hip_debug.so")
import torch
from caspr_amd import ops
from caspr_amd.models import CaSPR
from caspr_amd.utils.synthetic import seeded_state_dict, random_clouds, car_sequences
from oracle import model as O

dev = torch.device("cuda:0")
m = CaSPR()
sd = seeded_state_dict(m.state_dict(), 0)
m.load_state_dict(sd)
m = m.to(dev).eval()
sd64 = {k: v.double() for k, v in sd.items()}
le = m.encoder.local_extract
kind = sys.argv[sys.argv.index("--clouds") + 1] if "--clouds" in sys.argv else "random"
x = random_clouds(1, 2, 4096, seed=7) if kind == "random" else car_sequences(1, 2, 2048, seed=1234)[0]
torch.set_num_threads(32)
with torch.no_grad():
    xyz, feat = ops.prep_input(x.to(dev))
    idx = le.indices(xyz)
    torch.cuda.synchronize()
    p64 = xyz.cpu().double()
    f64 = torch.stack([p64[..., 0] ** 2, p64[..., 1] ** 2, p64[..., 2] ** 2, p64[..., 0] * p64[..., 2], p64[..., 0] * p64[..., 1], p64[..., 2] * p64[..., 1]], dim=-1)
    cur_xyz, cur_feat64, C = xyz, f64, 6
    for level in range(2):
        sa = le.set_abstractions[level]
        d = idx["sa"][level]
        ctr = d["new_xyz"]
        M = ctr.shape[1]
        B = cur_xyz.shape[0]
        outs64 = []
        lo_in = level == 1 and "--no-lo" not in sys.argv
        ldf = (C + 3) // 4 * 4 if not lo_in else 2 * C
        fin = torch.zeros(B, cur_xyz.shape[1], ldf, device=dev)
        hi = cur_feat64.float()
        fin[:, :, :C] = hi.to(dev)
        if lo_in:      # the previous level's output as hi + lo (ops.FEAT_LO_IN)
            fin[:, :, C:2 * C] = (cur_feat64 - hi.double()).float().to(dev)
        for s, ns in enumerate(sa.layers):
            bidx = d["ball_idx"][s]
            bi = bidx.cpu().long()
            K = torch.tensor([[len(set(r.tolist())) for r in fr] for fr in bi])                 # (B, M)
            # grouped input in f64: [p - centre (f32 subtraction, as the grouper) | features]
            pc, cc = cur_xyz.cpu(), ctr.cpu()
            gx = torch.stack([torch.gather(pc[b_], 0, bi[b_].reshape(-1, 1).expand(-1, 3)).view(M, ns, 3) - cc[b_].view(M, 1, 3) for b_ in range(B)])   # f32
            gf = torch.stack([torch.gather(cur_feat64[b_], 0, bi[b_].reshape(-1, 1).expand(-1, C)).view(M, ns, C) for b_ in range(B)])
            grouped = torch.cat([gx.double(), gf], dim=-1).permute(0, 1, 3, 2).reshape(B * M, C + 3, ns)
            pre = "encoder.local_extract.set_abstractions.%d.pointnet_modules.%d" % (level, s)
            want = O.feature_extractor(sd64, pre, grouped).view(B, M, -1)
            outs64.append(want)
            for rk in ("-1", "4", "8"):
                os.environ["CASPR_SA_REPAIR_K"] = rk
                Co = want.shape[2]
                out = torch.zeros(B, M, 2 * Co, device=dev)
                ops.sa_mlp_max(cur_xyz, ctr, fin, bidx, C, sa.pointnet_modules[s].kernel_layers(), out, 0,
                               feat_kind=(3 if level == 0 else 0) | ops.FEAT_LO_OUT | (ops.FEAT_LO_IN if lo_in else 0))
                err = (out[:, :, :Co].cpu().double() - want).abs().amax(dim=2)                    # (B, M): the f32 output
                err_hl = (out[:, :, :Co].cpu().double() + out[:, :, Co:].cpu().double() - want).abs().amax(dim=2)     # hi + lo
                line = []
                for k in sorted(set(K.view(-1).tolist())):
                    sel = K == k
                    line.append("K=%d: n %d max %.1e (hi+lo %.1e) bad %d" % (k, int(sel.sum()), float(err[sel].max()), float(err_hl[sel].max()), int((err[sel] > 1e-5).sum())))
                print("level %d scale %d (ns %d) repair<=%s: overall max %.2e | %s" % (level, s, ns, rk, float(err.max()), " | ".join(line)), flush=True)
        cur_feat64 = torch.cat(outs64, dim=2)
        C = cur_feat64.shape[2]
        cur_xyz = ctr
