#!/usr/bin/env python
"""Round 6, a committed negative: farthest-point sampling of the later levels on a side stream BESIDE a partner kernel on the main stream,
against the idle-chip result.  With the leaner FPS kernel of round 6 (packed distances, 32-bit selection; kept out of the tree) the
selections went wrong beside the global PointNet's stats-only conv; with the kernel in the tree they never do.  `--lib <name>.so` picks a
library variant next to libcaspr_hip.so."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from caspr_amd import lib
if "--lib" in sys.argv:
    lib.SO_PATH = lib.SO_PATH.replace("libcaspr_hip.so", sys.argv[sys.argv.index("--lib") + 1])
print("library:", os.path.basename(lib.SO_PATH))
from caspr_amd import ops
from caspr_amd.models import CaSPR
from caspr_amd.utils.synthetic import seeded_state_dict, car_sequences
dev = torch.device("cuda:0")
x, sp = car_sequences(16, 10, 2048, seed=1234)
m = CaSPR(); m.load_state_dict(seeded_state_dict(m.state_dict(), 0)); m = m.to(dev).eval()
xg = x.to(dev)
le = m.encoder.local_extract
xyz, feat = ops.prep_input(xg, True, True)
with torch.no_grad():
    ind = le.indices(xyz)
torch.cuda.synchronize()
sas = le.set_abstractions
kind = ops.FEAT_QUAD | ops.FEAT_PAIRS
x0 = ind["sa"][0]["new_xyz"]; x1 = ind["sa"][1]["new_xyz"]; x2 = ind["sa"][2]["new_xyz"]
ref = {n_: ops.furthest_point_sampling(c, M)[0:1][0].clone() if False else ops.furthest_point_sampling(c, M).clone() for n_, (c, M) in {"fps1": (x0, 512), "fps2": (x1, 256), "fps3": (x2, 64)}.items()}
torch.cuda.synchronize()
side = torch.cuda.Stream()
def partner(which):
    with torch.no_grad():
        if which == "sa0":
            return sas[0].run(xyz, feat, 6, None, ind["sa"][0], feat_kind=kind, lo_in=False, lo_out=True)
        if which == "global":
            B, T, N, _ = xg.shape
            X1 = torch.empty(B, T * N, 512 + 64, device=dev)
            return m.encoder.global_extract.features(xg.view(B, T * N, 4), y1_out=X1[:, :, 512:])
        if which == "ball":
            return ops.ball_query(0.05, 32, xyz, x0)
        if which == "three_nn":
            return ops.three_nn(xyz, x0, with_weights=True)
        if which == "prep":
            return ops.prep_input(xg, True, True)
        if which.startswith("conv"):
            L = LAY[which]
            if which == "conv4_64":
                return ops.conv1x1(L["pw"], L["bias"], L["x"])
            if which == "gn_stats":
                return ops.gn_stats(L["x"], 64, L["g"], L["be"])
            return ops.conv1x1_gn(L["pw"], L["bias"], L["x"], L["g"], L["be"], in_scale=L["sc"], in_shift=L["sh"], in_relu=True, want_max=True, write=(which != "conv128_1024"))
torch.manual_seed(1)
def mk(B, P, Cin, Cout):
    w = torch.randn(Cout, Cin, device=dev) / Cin ** 0.5
    return dict(pw=ops.PackedWeight(w), bias=torch.randn(Cout, device=dev), x=torch.randn(B, P, Cin, device=dev), sc=torch.rand(B, Cin, device=dev) + 0.5,
                sh=torch.randn(B, Cin, device=dev), g=torch.ones(Cout, device=dev), be=torch.zeros(Cout, device=dev))
LAY = {"conv4_64": mk(16, 20480, 4, 64), "conv64_128": mk(16, 20480, 64, 128), "conv128_1024": mk(16, 20480, 128, 1024), "conv512_512": mk(160, 1024, 512, 512)}
LAY["gn_stats"] = dict(x=torch.randn(16, 20480, 64, device=dev), g=torch.ones(64, device=dev), be=torch.zeros(64, device=dev))
def partner2(which):
    L = LAY[which]
    if which == "conv4_64": return ops.conv1x1(L["pw"], L["bias"], L["x"])
    if which == "gn_stats": return ops.gn_stats(L["x"], 64, L["g"], L["be"])
    return ops.conv1x1_gn(L["pw"], L["bias"], L["x"], L["g"], L["be"], in_scale=L["sc"], in_shift=L["sh"], in_relu=True, want_max=True, write=(which != "conv128_1024"))
_p = partner
def partner(which):
    return partner2(which) if which in LAY else _p(which)
for which in ("none", "conv128_1024"):
    bad = {k: [] for k in ref}
    for r in range(8):
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            got = {"fps1": ops.furthest_point_sampling(x0, 512), "fps2": ops.furthest_point_sampling(x1, 256), "fps3": ops.furthest_point_sampling(x2, 64)}
        if which != "none":
            keep = partner(which)
        torch.cuda.synchronize()
        for k in ref:
            bad[k].append(int((got[k] != ref[k]).sum()))
    print("FPS beside %-9s: wrong indices per run %s" % (which, bad), flush=True)
# anatomy of one failure: the first wrong round of the first wrong frame of fps2 beside conv128_1024
import numpy as np
for attempt in range(20):
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        g2 = ops.furthest_point_sampling(x1, 256)
    keep = partner("conv128_1024")
    torch.cuda.synchronize()
    dm = (g2 != ref["fps2"])
    if dm.any():
        f = int(dm.any(dim=1).nonzero()[0]); r0 = int(dm[f].nonzero()[0])
        c = x1[f].cpu().double().numpy(); want = ref["fps2"][f].cpu().numpy(); got = g2[f].cpu().numpy()
        # running minimum distance before round r0
        sel = want[:r0]
        dmin = np.min(((c[:, None, :] - c[sel][None, :, :]) ** 2).sum(-1), axis=1)
        order = np.argsort(-dmin)
        print("frame %d first wrong round %d: want %d (d2 %.6g), got %d (d2 %.6g); rank of got among candidates: %d; top-4 candidates %s; previous picks %s; got was picked before: %s; thread of want %d, of got %d" % (
            f, r0, want[r0], dmin[want[r0]], got[r0], dmin[got[r0]], int(np.where(order == got[r0])[0][0]), order[:4].tolist(), want[max(0, r0 - 3):r0].tolist(), bool(got[r0] in sel), want[r0] % 256, got[r0] % 256))
        print("next rounds want", want[r0:r0 + 6].tolist(), "got", got[r0:r0 + 6].tolist())
        break
