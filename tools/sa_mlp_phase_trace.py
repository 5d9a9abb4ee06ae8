#!/usr/bin/env python
"""Phase cycle breakdown of sa_mlp_kernel (thread 0 of one workgroup) via s_memtime stamps, for SA3..SA5."""
import ctypes, sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from caspr_amd import lib, ops
# phase traces / experiment switches live in the debug flavour only: CASPR_BUILD_DEBUG=1 python caspr_amd/csrc/build.py
lib.SO_PATH = lib.SO_PATH.replace("libcaspr_hip.so", "libcaspr_hip_debug.so")
from caspr_amd.models import CaSPR
from caspr_amd.utils.synthetic import seeded_state_dict, car_sequences
dev = torch.device("cuda:0")
m = CaSPR(); m.load_state_dict(seeded_state_dict(m.state_dict(), 0)); m = m.to(dev).eval()
so = ctypes.CDLL(lib.SO_PATH)
x, _ = car_sequences(16, 10, 2048)
xyz, feat = ops.prep_input(x.to(dev))
names = ["setup+gather", "L1 mfma", "L1 stats", "L1 norm", "L2 mfma", "L2 stats", "L2 norm", "L3 mfma", "L3 stats", "L3 max"]
sa = m.encoder.local_extract.set_abstractions
cur_xyz, cur_feat, C = xyz, feat, 6
for lvl in range(5):
    idx = sa[lvl].indices(cur_xyz)
    if lvl >= 2:
        for sc in range(2):
            out = torch.empty(cur_xyz.shape[0], sa[lvl].num_points_out, sa[lvl].get_num_features_out(), device=dev)
            buf = torch.zeros(16, dtype=torch.int64, device=dev)
            pn = sa[lvl].pointnet_modules[sc]
            layers = pn.kernel_layers()
            pre_mode = "--pre" in sys.argv and lvl < 4        # the production path of levels 3-4: first layer pre-aggregated (stamps 1-2: the gather fills layer 1's tile)
            if pre_mode:
                pw_f, wx = pn.pre_layers()
                pre = ops.conv1x1(pw_f, None, cur_feat)
                call = lambda: ops.sa_mlp_max_pre(cur_xyz, idx["new_xyz"], pre, idx["ball_idx"][sc], wx, layers, out, 0)
            else:
                call = lambda: ops.sa_mlp_max(cur_xyz, idx["new_xyz"], cur_feat, idx["ball_idx"][sc], C, layers, out, 0)
            call(); torch.cuda.synchronize()
            so.caspr_debug_set_sa_trace(ctypes.c_void_p(buf.data_ptr()))
            t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True); t0.record()
            call()
            t1.record(); torch.cuda.synchronize()
            so.caspr_debug_set_sa_trace(ctypes.c_void_p(0))
            t = buf.cpu()
            d = (t[1:11] - t[0:10]).tolist()
            print("SA%d scale %d: kernel %.3f ms; total %d cycles: " % (lvl + 1, sc, t0.elapsed_time(t1), int(t[10] - t[0])) + ", ".join("%s %d" % (n, v) for n, v in zip(names, d)))
    cur_xyz, cur_feat = sa[lvl].run(cur_xyz, cur_feat, C, idx=idx)
    C = cur_feat.shape[2]
